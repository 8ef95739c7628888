#!/usr/bin/env python3
"""A/B the flux-gather kernel variants on the bench workload (interleaved rounds in one
process, medians reported).  Diagnostic tool — not part of the product path."""
import argparse
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd"))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
import mgcfd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lattice", type=int, default=67)
ap.add_argument("--variants", type=str, default="1,2,3,4,5,6,7")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--launches", type=int, default=50)
args = ap.parse_args()

mg, levels = bench.build_workload(args.lattice)
s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
nel, n_int = s.nel(0), s.num_internal_edges(0)
s.set(0, "variables", bench.perturbed_state(nel, s.far_field()[:5]))
algo = 40 * n_int + 80 * nel
variants = [int(v) for v in args.variants.split(",")]
res = {}
for rnd in range(args.rounds):
    for exact in (1, 0):
        s.set_option("exact", exact)
        for v in variants:
            s.set_option("flux_variant", v)
            t = s.bench_flux(0, args.launches)
            res.setdefault((exact, v), []).append(t)
print(f"nodes {nel} internal edges {n_int}  algorithmic bytes/launch {algo}")
for (exact, v), ts in sorted(res.items()):
    med = statistics.median(ts)
    print(f"exact={exact} variant={v}: median {med*1e6:8.2f} us  min {min(ts)*1e6:8.2f} us  "
          f"{n_int/med/1e9:6.2f} Gedges/s  roofline frac {algo/med/8e12:.3f}")
