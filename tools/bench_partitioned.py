#!/usr/bin/env python3
"""BASELINE config 5 in its synthetic form: ONE level (default 134^3 = 2.4 M nodes / 7.2 M edges, the M6-L0-like level
tiled 8x) split over the ranks by recursive coordinate bisection, fused sweeps with a halo message after every
Runge-Kutta stage and one all-reduce(MIN) per sweep (strong scaling: the total work is fixed).

    python tools/bench_partitioned.py                                   # 1 GPU: the whole level, no ghosts
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_partitioned.py --gpus N
    MGCFD_BENCH_REHEARSAL=1 ... (every rank on device 0, gloo: a functional rehearsal on a one-GPU box, not a measurement)

Prints one JSON line on rank 0: whole-job Medges/s of compute_flux_edge, ms per sweep, halo volume."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--lattice", type=int, default=134)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--warmup", type=int, default=20)
a = ap.parse_args()
import torch
import bench, mgcfd
from mgcfd.distributed import HipSolverAdapter, PartitionedSweep
from mgcfd.partition import halo_volume, partition_level, rcb_partition
world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
rehearsal = os.environ.get("MGCFD_BENCH_REHEARSAL") == "1"
if rehearsal:
    local_rank = 0
torch.cuda.set_device(local_rank)
dist = None
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if rehearsal:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
mg, levels = bench.build_workload(a.lattice)                     # every rank builds the same level and the same partition
L = levels[0]
part = rcb_partition(np.asarray(L["coords"]), world)
P = partition_level(L, part)[rank]
dev = torch.device("cuda", local_rank)
st = torch.cuda.Stream(device=local_rank); torch.cuda.set_stream(st)
s = mgcfd.Solver.from_arrays([P.level], mg.mesh_variant, device=local_rank, n_owned=[P.n_owned])
s.set_stream(st.cuda_stream)
q = bench.perturbed_state(L["nel"], s.far_field()[:5])
s.set(0, "variables", q[P.global_ids])
sw = PartitionedSweep(HipSolverAdapter(s, dev), P, dist, make_buffer=lambda n: torch.empty(max(n, 1), dtype=torch.float64, device=dev), fused=True)
def barrier():
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
for _ in range(a.warmup):
    sw.sweep()
barrier(); t0 = time.perf_counter()
for _ in range(a.steps):
    sw.sweep()
barrier(); elapsed = time.perf_counter() - t0
t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
if world > 1:
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
rc, bad = s.check_for_invalid_variables(0)
if rc != 0:
    raise SystemExit(f"state became invalid (code {rc}, cell {bad})")
if rank == 0:
    n_int = int(L["n_internal"])
    print(json.dumps({"metric": "Medges/s (compute_flux_edge), one level partitioned over the ranks", "value": round(3 * n_int * a.steps / float(t.item()) / 1e6, 3),
                      "unit": "Medges/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(float(t.item()) / a.steps * 1e3, 6),
                      "scaling": "strong", "dtype": "f64", "data": "synthetic",
                      "config": {"workload": f"{a.lattice}^3 M6-L0-like level: {L['nel']} nodes / {n_int} internal edges in total, recursive coordinate bisection into {world} parts",
                                 "halo_nodes_total": int(halo_volume(L, part)) if world > 1 else 0, "rehearsal": rehearsal}}))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
