"""Is the torch.distributed path (PartitionedSweep over gloo, every rank on device 0: bench.py's rehearsal) itself reproducible?
N ranks, the level in N parts, SWEEPS sweeps by the torch path, against the whole level on each rank.
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/torch_path_check.py [sweeps]
(gloo's point-to-point calls on device tensors ignore streams: mgcfd/distributed.py stages such messages through the host;
before it did, 11 of 12 runs of this script differed from the whole level)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mg-cfd-app-plain_amd", "", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch, torch.distributed as dist
import bench, mgcfd
from mgcfd.distributed import HipSolverAdapter, PartitionedSweep
from mgcfd.partition import partition_level, rcb_partition
sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
stream = torch.cuda.Stream(device=0); torch.cuda.set_stream(stream)
mg, levels = bench.build_workload(30)
L = levels[0]
P = partition_level(L, rcb_partition(np.asarray(L["coords"]), world))[rank]
whole = mgcfd.Solver.from_arrays([L], mg.mesh_variant)
q = bench.perturbed_state(L["nel"], whole.far_field()[:5])
whole.set(0, "variables", q); whole.smooth(0, sweeps); want = whole.get(0, "variables"); whole.close()
s = mgcfd.Solver.from_arrays([P.level], mg.mesh_variant, n_owned=[P.n_owned]); s.set_stream(stream.cuda_stream)
s.set(0, "variables", q[P.global_ids])
sw = PartitionedSweep(HipSolverAdapter(s, dev), P, dist, make_buffer=lambda n: torch.empty(max(n, 1), dtype=torch.float64, device=dev), fused=True)
sw.exchange("variables")
for _ in range(sweeps): sw.sweep()
torch.cuda.synchronize()
got = s.get(0, "variables")
bad = int(np.count_nonzero(np.any(got.view(np.int64) != want[P.global_ids].view(np.int64), axis=1)))
print(f"rank {rank}: torch path after {sweeps} sweeps: {bad} node(s) differ from the whole level", flush=True)
dist.barrier(); s.close(); dist.destroy_process_group()
