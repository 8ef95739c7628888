#!/usr/bin/env python3
"""Bitwise comparison of the flux variants on the bench workload: fluxes of one launch and the
state after a few fused sweeps must be identical whichever variant computes them."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
import bench, mgcfd
lattice = int(sys.argv[1]) if len(sys.argv) > 1 else 67
mg, levels = bench.build_workload(lattice)
s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
q0 = bench.perturbed_state(s.nel(0), s.far_field()[:5])
out = {}
for v in (0, 1, 2, 3, 4):
    s.set_option("flux_variant", v)
    s.set(0, "variables", q0)
    s.zero_fluxes(0)
    s.compute_fluxes(0)
    f = s.get(0, "fluxes").copy()
    s.set(0, "variables", q0)
    for _ in range(3):
        s.smooth(0)
    out[v] = (f, s.get(0, "variables").copy())
for v in (1, 2, 3, 4):
    print("variant", v, "fluxes identical:", np.array_equal(out[0][0].view(np.int64), out[v][0].view(np.int64)),
          " after 3 sweeps identical:", np.array_equal(out[0][1].view(np.int64), out[v][1].view(np.int64)))
