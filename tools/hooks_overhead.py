#!/usr/bin/env python3
"""Host-side cost of the split sweep (mgcfd_sweep_begin / _flux0 / _end, what a multi-rank run calls
around its all-reduce) against the graph-replayed mgcfd_smooth, on one GPU without a collective."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
import torch
import bench, mgcfd
from mgcfd.distributed import HipSolverAdapter, ShardedSweep
mg, levels = bench.build_workload(67)
s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s.set_stream(st.cuda_stream)
s.set(0, "variables", bench.perturbed_state(s.nel(0), s.far_field()[:5]))
sw = ShardedSweep(HipSolverAdapter(s, torch.device("cuda", 0)), None)
sw.overlap_even_alone = True
def timeit(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
print("smooth              us/step", round(timeit(lambda: s.smooth(0, 1)), 1))
print("split sweep (hooks) us/step", round(timeit(lambda: sw.sweep(0)), 1))
t = s_min = HipSolverAdapter(s, torch.device("cuda", 0)).min_tensor(0)
def with_dummy_collective():
    s.sweep_begin(0); t.mul_(1.0); s.sweep_flux0(0); s.sweep_end(0)
print("split sweep + a torch op on the scalar us/step", round(timeit(with_dummy_collective), 1))
s.set_option("graph", 0)
print("smooth (eager launches) us/step", round(timeit(lambda: s.smooth(0, 1)), 1))
s.set_option("graph", 1)
print("smooth (graph) again    us/step", round(timeit(lambda: s.smooth(0, 1)), 1))
print("smooth(0, 10) per sweep (graph)", round(timeit(lambda: s.smooth(0, 10), 60) / 10, 1))
s.set_option("graph", 0)
print("smooth(0, 10) per sweep (eager)", round(timeit(lambda: s.smooth(0, 10), 60) / 10, 1))
