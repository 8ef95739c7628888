#!/usr/bin/env python3
"""What the partitioned V-cycle machinery costs a rank: the bench hierarchy as ONE rank of a group (no peers: every exchange
is an empty message, every wait a no-op) through mgcfd_group_cycles against mgcfd_run_cycles on the same hierarchy; and the
hierarchy in N parts on this one GPU (the ranks share it: GPU time adds up, the host issues N ranks' calls from one thread)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
import numpy as np
import bench, mgcfd
from mgcfd.partition import partition_hierarchy, rcb_partition
mg, levels = bench.build_hierarchy()
whole = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
whole.run_cycles(50)
t = time.perf_counter(); whole.run_cycles(50); t_whole = (time.perf_counter() - t) / 50
whole.close()
print(f"mgcfd_run_cycles, one solver: {t_whole * 1e3:.4f} ms per cycle")
for n in [int(x) for x in (sys.argv[1:] or ["1", "2", "4", "8"])]:
    H = partition_hierarchy(levels, rcb_partition(np.asarray(levels[0]["coords"]), n))
    solvers = []
    for h in H:
        lv, owned, keys = h.solver_args()
        solvers.append(mgcfd.Solver.from_arrays(lv, mg.mesh_variant, n_owned=owned, order_keys=keys))
    g = mgcfd.Group(solvers)
    for h, s in zip(H, solvers):
        for l in range(len(levels)):
            s.rank_set_halo(l, h.levels[l])
    for l in range(len(levels)):
        g.exchange(l)
    g.cycles(10, rms=False)
    t = time.perf_counter(); g.cycles(25, rms=False); dt = (time.perf_counter() - t) / 25
    ghosts = [sum(h.levels[l].level["nel"] - h.levels[l].n_owned for h in H) for l in range(len(levels))]
    print(f"mgcfd_group_cycles, {n} rank(s) sharing this GPU: {dt * 1e3:.4f} ms per cycle end to end; ghost nodes per level (all ranks) {ghosts}")
    g.close()
    for s in solvers:
        s.close()
