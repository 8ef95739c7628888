#!/usr/bin/env python3
"""The flux kernel, its probe and the fused sweep on the mixed-element level (bench.py --mesh mixed), exact / contracted / order-free."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
import bench, mgcfd
def make_solver(levels, variant):
    return mgcfd.Solver.from_arrays(levels, variant)
print(json.dumps(bench.mixed_mesh_roofline(make_solver, False), indent=1))
mg, levels = bench.build_workload(67, mesh="mixed")
s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
s.set(0, "variables", bench.perturbed_state(s.nel(0), s.far_field()[:5]))
print("half rows:", s.has_half_rows(0), "order-free plan:", s.has_order_free(0), "edge once:", s.has_edge_once(0))
for name, exact, v in (("exact", 1, -1), ("contracted", 0, 1), ("order-free", 0, 65)):
    s.set_option("exact", exact); s.set_option("flux_variant", v)
    s.bench_flux(0, 500)
    print(f"{name:11s} flux launch {s.bench_flux(0, 1000) * 1e6:7.2f} us")
