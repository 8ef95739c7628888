#!/usr/bin/env python3
"""How much of a sweep's time is idle chip (launch boundaries, the tails of 1.5-round launches)?  An upper bound, measured: two
solvers on the same level sweeping at the same time on two streams — independent launches the hardware may overlap freely —
against one solver alone.  What the pair gains over two runs in a row is what ANY scheme that fills the tails (a dataflow
grid across Runge-Kutta stages) could gain at most.   python tools/exp/two_streams.py [sweeps=1000] [fast=0]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
import bench, mgcfd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
fast = len(sys.argv) > 2 and sys.argv[2] == "1"
mg, levels = bench.build_workload(67)
solvers = [mgcfd.Solver.from_arrays(levels, mg.mesh_variant) for _ in range(2)]
for s in solvers:
    s.set_option("exact", 0 if fast else 1)
    s.set(0, "variables", bench.perturbed_state(s.nel(0), s.far_field()[:5]))
    s.smooth(0, 300); s.synchronize()
def alone(s):
    t0 = time.perf_counter(); s.smooth(0, n); s.synchronize(); return (time.perf_counter() - t0) / n
a = min(alone(solvers[0]) for _ in range(3))
def both():
    th = [threading.Thread(target=lambda s=s: (s.smooth(0, n), s.synchronize())) for s in solvers]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return (time.perf_counter() - t0) / n
b = min(both() for _ in range(3))
print(f"{'fast' if fast else 'bit-identical'}: one solver {a * 1e6:.2f} us per sweep; two solvers at once {b * 1e6:.2f} us per pair of sweeps = {b / 2 * 1e6:.2f} us per sweep ({2 * a / b:.3f} x two in a row)")
# ... and the standalone compute_flux_edge launch (what roofline.frac prices): one solver's batch of launches alone, two solvers'
# batches at once on two streams; bytes = 40 E + 80 N per launch
algo = 40 * solvers[0].num_internal_edges(0) + 80 * solvers[0].nel(0)
for name, ex, var in (("bit-identical", 1, 1), ("order-free", 0, 65)):
    for s in solvers:
        s.set_option("exact", ex); s.set_option("flux_variant", var); s.bench_flux(0, 500)
    one = min(solvers[0].bench_flux(0, 1000) for _ in range(3))
    out = [0.0, 0.0]
    def run(k):
        out[k] = solvers[k].bench_flux(0, 1000)
    best = 1e9
    for _ in range(3):
        th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        best = min(best, (time.perf_counter() - t0) / 2000)
    print(f"flux launch, {name}: alone {one * 1e6:.2f} us = {algo / one / 8e12:.3f} of 8 TB/s; two batches at once: {best * 1e6:.2f} us per launch (wall / 2,000 launches) = {algo / best / 8e12:.3f}; "
          f"each batch's own event time {out[0] * 1e6:.2f} / {out[1] * 1e6:.2f} us per launch")

