#!/usr/bin/env python3
"""What the C++-side partitioned sweep (mgcfd_group_sweeps) costs a rank beyond the plain fused sweep, on one GPU:
(a) the whole bench level as a 1-rank group (no peers: min reduction + 8-byte copy + the same three stage launches),
(b) the level cut in two, ONE half timed as a rank whose peer is silent (its ghosts never change): boundary tiles, pack,
    device-to-device message to itself-as-peer is not possible, so (b) runs both halves on this GPU and reports the
    pair's time per sweep next to twice a half-size level's plain sweep.
    python tools/coupling_cost2.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
import bench, mgcfd
from mgcfd.partition import partition_level, rcb_partition

def timed(fn, sync, n=300, warm=30):
    for _ in range(warm): fn()
    sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    sync(); return (time.perf_counter() - t0) / n * 1e6

mg, levels = bench.build_workload(67)
L = levels[0]
s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
q = bench.perturbed_state(s.nel(0), s.far_field()[:5])
s.set(0, "variables", q)
print(f"plain fused sweep (mgcfd_smooth): {timed(lambda: s.smooth(0, 1), s.synchronize):.1f} us")
s.close()
for n_parts in (1, 2, 4):
    parts = partition_level(L, rcb_partition(np.asarray(L["coords"]), n_parts))
    solvers = []
    for P in parts:
        r = mgcfd.Solver.from_arrays([P.level], mg.mesh_variant, n_owned=[P.n_owned])
        r.set(0, "variables", q[P.global_ids])
        solvers.append(r)
    g = mgcfd.Group(solvers)
    for P, r in zip(parts, solvers):
        r.rank_set_halo(0, P)
    g.exchange(0)
    info = [r.rank_halo_info(0) for r in solvers]
    t = timed(lambda: g.sweeps(0, 1), g.synchronize)
    print(f"{n_parts} rank(s) on this one GPU, library loop: {t:.1f} us per sweep of the whole level; rank 0: {info[0]}")
    g.close()
    for r in solvers: r.close()
