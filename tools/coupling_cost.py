#!/usr/bin/env python3
"""What a rank pays for the per-sweep coupling of the mesh-copy sharding, without any wire time: the split sweep
around an all-reduce on a ONE-rank RCCL group against the plain sweep, on one GPU (host enqueue time and wall time)."""
import os, sys, time, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import bench, mgcfd
from mgcfd.distributed import HipSolverAdapter, ShardedSweep
sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
mg, levels = bench.build_workload(67)
s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s.set_stream(st.cuda_stream)
s.set(0, "variables", bench.perturbed_state(s.nel(0), s.far_field()[:5]))
sw = ShardedSweep(HipSolverAdapter(s, torch.device("cuda", 0)), None)
sw.dist = dist                      # force the collective path with a one-rank group
def timeit(fn, n=1000):
    for _ in range(100): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t) / n * 1e6, (t2 - t) / n * 1e6
for rp in (True, False):
    sw.reduce_partials = rp
    h, g = timeit(lambda: sw.sweep(0))
    print(f"split sweep around all_reduce(MIN) of the {'partial minima' if rp else 'scalar'}: host enqueue {h:.1f} us/step, wall {g:.1f} us/step")
h, g = timeit(lambda: s.smooth(0, 1))
print(f"plain sweep: host enqueue {h:.1f} us/step, wall {g:.1f} us/step")
dist.destroy_process_group()
