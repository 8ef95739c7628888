#!/usr/bin/env python3
"""What the RCCL rank loop costs the HOST per sweep and per V-cycle: the bench level / hierarchy as the ONE rank of an RCCL
communicator (no peers: every message is empty, but every launch, event and collective call of the loop is issued) through
mgcfd_rank_sweeps / mgcfd_rank_cycles, host time until the calls return and until the GPU is done, beside the plain fused
sweep / cycle.   python tools/hostcost_rccl.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
import bench, mgcfd
from mgcfd.partition import partition_hierarchy, partition_level, rcb_partition
mg, levels = bench.build_workload(67)
L = levels[0]
P = partition_level(L, rcb_partition(np.asarray(L["coords"]), 1))[0]
s = mgcfd.Solver.from_arrays([P.level], mg.mesh_variant, n_owned=[P.n_owned])
q = bench.perturbed_state(L["nel"], s.far_field()[:5])
s.set(0, "variables", q[P.global_ids])
s.smooth(0, 300); s.synchronize()
t0 = time.perf_counter(); s.smooth(0, 300); t1 = time.perf_counter(); s.synchronize(); t2 = time.perf_counter()
print(f"plain fused sweeps: host {(t1 - t0) / 300 * 1e6:.1f} us per sweep, until done {(t2 - t0) / 300 * 1e6:.1f}")
uid = mgcfd.rccl_unique_id()
s.rank_attach_rccl(0, 1, uid)
s.rank_set_halo(0, P)
s.rank_exchange(0)
for graph in (0, 1):
    s.set_option("graph", graph)
    s.rank_sweeps(0, 60); s.synchronize()
    t0 = time.perf_counter(); s.rank_sweeps(0, 300); t1 = time.perf_counter(); s.synchronize(); t2 = time.perf_counter()
    print(f"mgcfd_rank_sweeps, one RCCL rank{' (hipGraph replay)' if graph else ''}: host {(t1 - t0) / 300 * 1e6:.1f} us per sweep, until done {(t2 - t0) / 300 * 1e6:.1f}")
s.set_option("graph", 0)
s.rank_detach(); s.close()
mg, levels = bench.build_hierarchy()
whole = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
whole.run_cycles(30)
t = time.perf_counter(); whole.run_cycles(50); print(f"mgcfd_run_cycles: {(time.perf_counter() - t) / 50 * 1e3:.4f} ms per cycle")
whole.close()
H = partition_hierarchy(levels, rcb_partition(np.asarray(levels[0]["coords"]), 1))[0]
lv, owned, keys = H.solver_args()
s = mgcfd.Solver.from_arrays(lv, mg.mesh_variant, n_owned=owned, order_keys=keys)
s.rank_attach_rccl(0, 1, mgcfd.rccl_unique_id())
for l in range(len(lv)):
    s.rank_set_halo(l, H.levels[l])
for l in range(len(lv)):
    s.rank_exchange(l)
s.rank_cycles(10, rms=False)
t = time.perf_counter(); s.rank_cycles(50, rms=False); dt = (time.perf_counter() - t) / 50
print(f"mgcfd_rank_cycles, one RCCL rank: {dt * 1e3:.4f} ms per cycle end to end")
s.rank_detach(); s.close()
