"""One multigrid level per solver on the input of a directory (two solvers on this GPU, the Python loop) against run_cycles."""
import os, queue, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("mg-cfd-app-plain_amd", "", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch, mgcfd
from mgcfd.distributed import HipSolverAdapter, LevelPerRankCycle
d, cycles, world = sys.argv[1], int(sys.argv[2]), 2
dev = torch.device("cuda", 0)
whole = mgcfd.Solver.from_mesh(mgcfd.Mesh("input.dat", d))
rc = None
try:
    whole.run_cycles(cycles)
except mgcfd.MgcfdError as e:
    rc = e
print("whole:", rc, [bool(np.isfinite(whole.get(l, "variables")).all()) for l in range(whole.num_levels)], "check", [whole.check_for_invalid_variables(l) for l in range(whole.num_levels)])
want = [whole.get(l, "variables") for l in range(whole.num_levels)]
n_levels = whole.num_levels
tstream = torch.cuda.Stream()
solvers = [mgcfd.Solver.from_mesh(mgcfd.Mesh("input.dat", d)) for _ in range(world)]
for s in solvers: s.set_stream(tstream.cuda_stream)
boxes = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
def run(rank):
    torch.cuda.set_device(0); torch.cuda.set_stream(tstream)
    send = lambda t, dst: boxes[(rank, dst)].put(t.clone())
    recv = lambda t, src: t.copy_(boxes[(src, rank)].get(timeout=60))
    cyc = LevelPerRankCycle(HipSolverAdapter(solvers[rank], dev), n_levels, rank, world, send=send, recv=recv)
    for _ in range(cycles): cyc.cycle()
ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in ts]; [t.join() for t in ts]
for l in range(n_levels):
    s = solvers[l % world]
    got = s.get(l, "variables")
    print("level", l, "equal", np.array_equal(got.view(np.int64), want[l].view(np.int64)), "finite", bool(np.isfinite(got).all()), "check", s.check_for_invalid_variables(l))
