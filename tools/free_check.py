#!/usr/bin/env python3
"""Order-free flux kernel (fast namespace, variant 64) on the bench level: difference from the bit-identical kernel
after one launch and after a few fused sweeps, and event-timed batches of both.  Diagnostic tool."""
import os, sys, statistics
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
import bench, mgcfd
lattice = int(sys.argv[1]) if len(sys.argv) > 1 else 67
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 500
mg, levels = bench.build_workload(lattice)
s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
nel, n_int = s.nel(0), s.num_internal_edges(0)
print("half rows available:", s.has_half_rows(0))
q0 = bench.perturbed_state(nel, s.far_field()[:5])
out = {}
for name, exact, v in (("exact", 1, 1), ("contracted", 0, 1), ("free", 0, 64 | 1)):
    s.set_option("exact", exact); s.set_option("flux_variant", v)
    s.set(0, "variables", q0); s.zero_fluxes(0); s.compute_fluxes(0)
    f = s.get(0, "fluxes").copy()
    s.set(0, "variables", q0)
    for _ in range(3): s.smooth(0)
    out[name] = (f, s.get(0, "variables").copy())
for name in ("contracted", "free"):
    for k, what in ((0, "fluxes of one launch"), (1, "state after 3 sweeps")):
        a, b = out[name][k], out["exact"][k]
        print(f"{name:10s} {what}: max |diff| / max |ref| = {np.abs(a - b).max() / np.abs(b).max():.3e}")
algo = 40 * n_int + 80 * nel
res = {}
s.set(0, "variables", q0)
s.set_option("exact", 0); s.set_option("flux_variant", 65); s.bench_flux(0, 2000)      # clocks up
for rnd in range(5):
    for name, exact, v in (("exact", 1, 1), ("contracted", 0, 1), ("half", 0, 33), ("free", 0, 65)):
        s.set_option("exact", exact); s.set_option("flux_variant", v)
        res.setdefault(name, []).append(s.bench_flux(0, launches))
for name, ts in res.items():
    med = statistics.median(ts)
    print(f"{name:10s} median {med*1e6:7.2f} us  min {min(ts)*1e6:7.2f} us  {n_int/med/1e9:6.2f} Gedges/s  frac {algo/med/8e12:.3f}")
