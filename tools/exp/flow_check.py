#!/usr/bin/env python3
"""The dataflow sweep (MGCFD_FLOW=1: the three stages of a sweep as one launch, kernels.hip: k_sweep_flow) against the three
stage launches: bit for bit after N sweeps, and the time per sweep of both.   python tools/exp/flow_check.py [lattice=67] [sweeps=50] [mesh=lattice]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
import numpy as np
import bench, mgcfd
lattice = int(sys.argv[1]) if len(sys.argv) > 1 else 67
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
mesh = sys.argv[3] if len(sys.argv) > 3 else "lattice"
mg, levels = bench.build_workload(lattice, mesh=mesh)
res = {}
for flow in ("0", "1"):
    os.environ["MGCFD_FLOW"] = flow
    s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    q0 = bench.perturbed_state(s.nel(0), s.far_field()[:5])
    s.set(0, "variables", q0)
    s.smooth(0, n); s.synchronize()
    res[flow] = (s.get(0, "variables"), s.get(0, "residuals") if False else None, s.loop_iters(0)["flux"])
    s.smooth(0, 300); s.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); s.smooth(0, 1000); s.synchronize(); best = min(best, (time.perf_counter() - t0) / 1000)
    print(f"MGCFD_FLOW={flow}: {best * 1e6:.2f} us per sweep; state valid: {s.check_for_invalid_variables(0)[0] == 0}", flush=True)
    s.close()
a, b = res["0"][0], res["1"][0]
print(f"after {n} sweeps: nodes that differ {int(np.count_nonzero(np.any(a.view(np.int64) != b.view(np.int64), axis=1)))} of {len(a)}; flux iterations {res['0'][2]} / {res['1'][2]}")
