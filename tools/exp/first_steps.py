"""How long do the first sweeps after a synchronisation take?  (bench.py --steps 20 reads 68-70 us per step, --steps 2000 61.)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "mg-cfd-app-plain_amd"))
import numpy as np, torch
import bench, mgcfd
mg, levels = bench.build_workload(bench.LATTICE)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant, device=0); s.set_stream(stream.cuda_stream)
s.set(0, "variables", bench.perturbed_state(s.nel(0), s.far_field()[:5]))
for trial in range(3):
    for _ in range(5): s.smooth(0, 1)
    torch.cuda.synchronize()
    if trial == 2: time.sleep(0.5)
    n = 40
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    t0 = time.perf_counter()
    ev[0].record(stream)
    for k in range(n):
        s.smooth(0, 1); ev[k + 1].record(stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    d = [ev[k].elapsed_time(ev[k + 1]) * 1e3 for k in range(n)]
    print(f"trial {trial}: enqueue {1e6*(t1-t0)/n:.1f} us/step, wall {1e6*(t2-t0)/n:.1f} us/step; per-step GPU us:", " ".join(f"{x:.0f}" for x in d))
# one pair around K steps, K = 20, 100, 500
for K in (20, 20, 100, 500, 20):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record(stream)
    for _ in range(K): s.smooth(0, 1)
    b.record(stream); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"K={K}: events {a.elapsed_time(b)*1e3/K:.2f} us/step, wall {1e6*(t2-t0)/K:.2f} us/step")
