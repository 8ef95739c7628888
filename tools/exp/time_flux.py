#!/usr/bin/env python3
"""Event-timed batches of the standalone flux launch for the library MGCFD_LIB names: bit-identical (variant 1), contracted
(variant 1, exact 0) and order-free (variant 65) on one level.   python tools/exp/time_flux.py [lattice] [launches] [modes]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
import bench, mgcfd
lattice = int(sys.argv[1]) if len(sys.argv) > 1 else 67
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 500
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["exact", "free"]
mesh = os.environ.get("MGCFD_EXP_MESH", "lattice")
mg, levels = bench.build_workload(lattice, mesh=mesh)
s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
nel, n_int = s.nel(0), s.num_internal_edges(0)
s.set(0, "variables", bench.perturbed_state(nel, s.far_field()[:5]))
algo = 40 * n_int + 80 * nel
table = {"exact": (1, 1), "exactk": (1, 0), "contracted": (0, 1), "half": (0, 33), "free": (0, 65), "exacthalf": (1, 33), "exactonce": (1, 3), "exactidx": (1, 17)}
s.set_option("exact", 0); s.set_option("flux_variant", 65); s.bench_flux(0, 2000)      # clocks up
res = {}
for rnd in range(5):
    for name in modes:
        ex, v = table[name]
        s.set_option("exact", ex); s.set_option("flux_variant", v)
        res.setdefault(name, []).append(s.bench_flux(0, launches))
tag = os.path.basename(os.environ.get("MGCFD_LIB", "default")).replace("libmgcfd_hip_", "").replace(".so", "")
for name, ts in res.items():
    med = statistics.median(ts)
    print(f"{tag:14s} {mesh} {lattice} {name:10s} median {med*1e6:7.2f} us  min {min(ts)*1e6:7.2f} us  frac {algo/med/8e12:.3f}", flush=True)
