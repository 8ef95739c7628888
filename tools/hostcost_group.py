import os, sys, time
import numpy as np
ROOT = "/root/repo" if os.path.isdir("/root/repo/mg-cfd-app-plain_amd") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
import bench, mgcfd
from mgcfd.partition import partition_level, rcb_partition
mg, levels = bench.build_workload(67)
L = levels[0]
q = None
for n_parts in (1, 2):
    parts = partition_level(L, rcb_partition(np.asarray(L["coords"]), n_parts))
    solvers = []
    for P in parts:
        r = mgcfd.Solver.from_arrays([P.level], mg.mesh_variant, n_owned=[P.n_owned])
        if q is None: q = bench.perturbed_state(L["nel"], r.far_field()[:5])
        r.set(0, "variables", q[P.global_ids]); solvers.append(r)
    g = mgcfd.Group(solvers)
    for P, r in zip(parts, solvers): r.rank_set_halo(0, P)
    g.exchange(0); g.sweeps(0, 30); g.synchronize()
    t0 = time.perf_counter(); g.sweeps(0, 300); t1 = time.perf_counter(); g.synchronize(); t2 = time.perf_counter()
    print(f"{n_parts} ranks: host enqueue {(t1-t0)/300*1e6:.1f} us per sweep, until done {(t2-t0)/300*1e6:.1f} us per sweep")
    g.close(); [r.close() for r in solvers]
