#!/usr/bin/env python3
"""Sweeps (three fused stages each) on the bench level and V-cycles on the 4-level hierarchy: bit-identical, contracted and
order-free (fast namespace, variant 65)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
import numpy as np
import bench, mgcfd
mg, levels = bench.build_workload(67)
s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
q0 = bench.perturbed_state(s.nel(0), s.far_field()[:5])
for name, exact, v in (("exact", 1, -1), ("contracted", 0, -1), ("free", 0, 65)):
    s.set_option("exact", exact); s.set_option("flux_variant", v)
    s.set(0, "variables", q0)
    s.smooth(0, 2000); s.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); s.smooth(0, 2000); s.synchronize(); best = min(best, time.perf_counter() - t)
    print(f"sweep {name:10s}: {best / 2000 * 1e6:7.2f} us per sweep")
s.close()
mg, levels = bench.build_hierarchy()
s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
print("half rows per level:", [s.has_half_rows(l) for l in range(s.num_levels)])
ref = None
for name, exact, v in (("exact", 1, -1), ("contracted", 0, -1), ("free", 0, 65)):
    s2 = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    s2.set_option("exact", exact); s2.set_option("flux_variant", v)
    s2.run_cycles(25)
    q = s2.get(0, "variables").copy()
    if ref is None: ref = q
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); s2.run_cycles(25); best = min(best, time.perf_counter() - t)
    print(f"vcycle {name:10s}: {best / 25 * 1e3:7.4f} ms per cycle; level-0 variables after 25 cycles vs exact: max |diff|/max|ref| = {np.abs(q - ref).max() / np.abs(ref).max():.3e}")
    s2.close()
