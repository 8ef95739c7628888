#!/usr/bin/env python3
"""Sweeps on the bench level and V-cycles on the 4-level hierarchy for the library MGCFD_LIB names, both numerics modes.
   python tools/exp/time_sweep.py [sweeps=2000] [cycles=25] [modes=exact,fast]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
import bench, mgcfd
from mgcfd import meshgen
n_sw = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n_cy = int(sys.argv[2]) if len(sys.argv) > 2 else 25
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["exact", "fast"]
tag = os.path.basename(os.environ.get("MGCFD_LIB", "default")).replace("libmgcfd_hip_", "").replace(".so", "")
mg, levels = bench.build_workload(67)
s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
q0 = bench.perturbed_state(s.nel(0), s.far_field()[:5])
variant = int(os.environ.get("MGCFD_EXP_VARIANT", "-1"))
for mode in modes:
    s.set_option("exact", 1 if mode == "exact" else 0)
    s.set_option("flux_variant", variant)
    s.set(0, "variables", q0)
    s.smooth(0, 500); s.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); s.smooth(0, n_sw); s.synchronize(); best = min(best, (time.perf_counter() - t0) / n_sw)
    print(f"{tag:14s} variant {variant:3d} sweep  {mode:6s} {best * 1e6:7.2f} us", flush=True)
s.close()
if n_cy > 0:
    mg4 = meshgen.make_multigrid((67, 55, 48, 43), "m6wing", seed=0, jitter=0.2, area_noise=0.02, volume_noise=0.02)
    s = mgcfd.Solver.from_arrays(mgcfd.generated_to_levels(mg4), mg4.mesh_variant)
    for mode in modes:
        s.set_option("exact", 1 if mode == "exact" else 0)
        s.set_option("flux_variant", variant)
        s.run_cycles(4)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); s.run_cycles(n_cy); best = min(best, (time.perf_counter() - t0) / n_cy)
        print(f"{tag:14s} variant {variant:3d} vcycle {mode:6s} {best * 1e3:7.4f} ms", flush=True)
