#!/usr/bin/env python3
"""Event-timed batches of the order-free flux kernel from experiment builds (tools/exp_build.py): MGCFD_LIB=... python tools/exp/free_ablate.py"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
import bench, mgcfd
mg, levels = bench.build_workload(int(sys.argv[1]) if len(sys.argv) > 1 else 67)
s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
s.set(0, "variables", bench.perturbed_state(s.nel(0), s.far_field()[:5]))
s.set_option("exact", 0); s.set_option("flux_variant", 65)
s.bench_flux(0, 2000)
ts = [s.bench_flux(0, 500) for _ in range(5)]
print(os.environ.get("MGCFD_LIB", "default build"), f"free: median {statistics.median(ts)*1e6:.2f} us")
