#!/usr/bin/env python3
"""Time of the UNFUSED sweep (one launch per loop: flux, time_step, ... — what the drop-in's per-loop timers run) on the bench
level, for A/B runs of library variants (MGCFD_LIB).   python tools/exp/unfused_sweep_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
import bench, mgcfd
mg, levels = bench.build_workload(67)
s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
s.set(0, "variables", bench.perturbed_state(s.nel(0), s.far_field()[:5]))
s.set_option("fuse_update", 0)
s.smooth(0, 300); s.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); s.smooth(0, 500); s.synchronize(); best = min(best, (time.perf_counter() - t0) / 500)
print(f"unfused sweep: {best * 1e6:.2f} us  rms {s.calc_rms(0):.9e}")
