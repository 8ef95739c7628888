#!/usr/bin/env python3
"""bench.py — MG-CFD hot path on MI355X: Medges/s of the edge-flux sweep.

Contract (one JSON line on rank 0):
  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): "Onera-M6 L0 only, flux+update kernels only, no MG".
The dataset release is not available, so the M6-L0-like synthetic mesh of SURVEY.md §8d cfg2
is used: a 67^3 jittered lattice (300,763 nodes / 888,822 internal edges) with randomly
permuted node ids, mesh_name = m6wing (adjust + dampen 5e-8), state = far field with +-1 %
uniform noise (seed 1234).  One STEP = one smoothing sweep on that level, exactly the
per-level body of the reference's cycle loop (src/euler3d_cpu_double.cpp:383-508):
  copy old <- variables; compute_step_factor; 3 x [compute_flux_edge + boundary + wall fluxes,
  time_step]; residual.
value = internal edges pushed through compute_flux_edge (3 per step per rank) / wall time of
the K timed steps, max over ranks, summed over ranks (weak scaling: every rank owns one mesh
copy, coupled through the global-min time step exactly like the reference's -m duplication:
one all-reduce(MIN) of one fp64 per sweep over RCCL).

roofline: the dominant kernel, k_flux_tile — in the sweep it is launched as one whole
Runge-Kutta stage (fluxes of all three edge classes + time_step), so one launch carries the
ALGORITHMIC bytes of both loops it replaces: 40*E + 80*N (compute_flux_edge) + 168*N (time_step)
(SURVEY.md §8d).  Its mean duration is measured with hipEvent pairs on the launch stream during
the timed region (every 8th sweep runs eagerly with one pair around its three stage launches;
the other sweeps replay a hipGraph), against 8 TB/s.  `frac_if_priced_as_flux_only` and
`flux_kernel_alone` (the standalone compute_flux_edge kernel, 50 back-to-back launches) are
given beside it.
cpu_baseline: the reference's own compute_flux_edge (oracle/_ref, built from the reference
sources) — or the C oracle port when that build is absent — timed on one host core on the
same mesh for a bounded number of passes.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd"))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
LATTICE = 67                 # 67^3 = 300,763 nodes, 888,822 internal edges


def build_workload(lattice: int, seed: int = 0):
    from mgcfd import meshgen, generated_to_levels
    mg = meshgen.make_multigrid((lattice,), "m6wing", seed=seed, jitter=0.2, area_noise=0.02,
                                volume_noise=0.02, permute=True)
    return mg, generated_to_levels(mg)


def perturbed_state(nel, ff_var, seed=1234, amplitude=0.01):
    rng = np.random.default_rng(seed)
    base = np.tile(np.asarray(ff_var, dtype=np.float64), (nel, 1))
    return base * (1.0 + amplitude * rng.uniform(-1.0, 1.0, base.shape))


def cpu_baseline(levels, sample_seconds: float):
    """Time compute_flux_edge on ONE host core for a bounded number of passes."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    L = levels[0]
    edges = np.ascontiguousarray(L["edges"]).copy()
    # same edge-weight preconditioning the solver applies (validation.cpp:28-75)
    lib = O.load(native=True)
    coords = np.ascontiguousarray(L["coords"], dtype=np.float64)
    lib.ora_adjust_ewt(O.ptr(coords), len(edges), O.ptr(edges))
    lib.ora_dampen_ewt(len(edges), O.ptr(edges), 5e-8)
    ff = O.farfield()
    q = perturbed_state(L["nel"], ff.var)
    f = np.zeros_like(q)
    n_int = int(L["n_internal"])
    if O.have_reference():
        ref = O.load_reference()
        ref.ref_init(1, 2)
        fn = lambda: ref.ref_compute_flux_edge(0, n_int, O.ptr(edges), O.ptr(q), O.ptr(f))
        kind = "reference"
        how = "reference compute_flux_edge (src/Kernels/flux_loops.cpp:78-153) built by oracle/build_ref.sh, g++ -O3 -fno-fast-math -ffp-contract=off"
    else:
        fn = lambda: lib.ora_compute_flux_edge(0, n_int, O.ptr(edges), O.ptr(q), O.ptr(f))
        kind = "port"
        how = "oracle/mgcfd_oracle.c ora_compute_flux_edge, gcc -O3 -fno-fast-math -march=native"
    fn()
    t0 = time.perf_counter()
    fn()
    one = time.perf_counter() - t0
    passes = max(3, int(sample_seconds / max(one, 1e-6)))
    t0 = time.perf_counter()
    for _ in range(passes):
        fn()
    dt = time.perf_counter() - t0
    out = {"value": round(n_int * passes / dt / 1e6, 3), "unit": "Medges/s", "cores": 1, "kind": kind,
           "sample": f"{passes} passes of compute_flux_edge over the same {n_int}-edge level ({dt:.1f} s), {how}"}
    # ... and on all the host cores this process may use, the reference's own race-free way to use threads: one
    # private copy of the mesh state per thread (its -m duplication, src/Base/io_enhanced.cpp:89-201).  The oracle's
    # restatement is used here (same arithmetic, no global counters to race on); a reported baseline like the first.
    try:
        import threading
        cores = max(1, min(len(os.sched_getaffinity(0)), 16))     # a one-GPU box's CPU share is 16 cores
        per = max(2, int(sample_seconds / 4.0 / max(one, 1e-6)))
        state = [(q.copy(), np.zeros_like(q)) for _ in range(cores)]

        def work(k):
            qk, fk = state[k]
            for _ in range(per):
                lib.ora_compute_flux_edge(0, n_int, O.ptr(edges), O.ptr(qk), O.ptr(fk))

        threads = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
        t0 = time.perf_counter()
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        dt_all = time.perf_counter() - t0
        out["all_cores"] = {"value": round(n_int * per * cores / dt_all / 1e6, 3), "unit": "Medges/s", "cores": cores, "kind": "port",
                            "sample": f"{cores} threads x {per} passes, one private copy of the level's state per thread ({dt_all:.1f} s), "
                                      "oracle/mgcfd_oracle.c ora_compute_flux_edge, gcc -O3 -fno-fast-math -march=native"}
    except Exception as e:                                   # the single-core figure stands on its own
        out["all_cores"] = {"error": str(e)}
    return out


def vcycle_wall(fast: bool, cycles: int = 25):
    """Second half of BASELINE.json's metric: wall seconds per MG V-cycle on the 4-level M6-like hierarchy
    (67^3/55^3/48^3/43^3 lattices = 300,763/166,375/110,592/79,507 nodes, nearest-node maps,
    mesh_name = m6wing), 25 cycles as the reference's default (src/Base/config.cpp:63), best of 3."""
    import mgcfd
    from mgcfd import meshgen
    mg = meshgen.make_multigrid((67, 55, 48, 43), "m6wing", seed=0, jitter=0.2, area_noise=0.02, volume_noise=0.02)
    s = mgcfd.Solver.from_arrays(mgcfd.generated_to_levels(mg), mg.mesh_variant)
    s.set_option("exact", 0 if fast else 1)
    s.run_cycles(2)
    best = float("inf")
    for _ in range(3):
        s.reset_monitoring()
        t0 = time.perf_counter()
        rms = s.run_cycles(cycles)
        best = min(best, time.perf_counter() - t0)
    edge_iters = sum(s.loop_iters(l)["flux"] for l in range(s.num_levels)) // cycles
    out = {"workload": f"4-level M6-like synthetic hierarchy {[l.nel for l in mg.levels]} nodes, {cycles} cycles",
           "wall_s_per_cycle": round(best / cycles, 9), "flux_edge_iterations_per_cycle": edge_iters,
           "medges_per_s_whole_cycle": round(edge_iters * cycles / best / 1e6, 1), "rms_last": float(rms[-1])}
    s.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)        # 0.15 s of GPU time: past the clock ramp of the first ms
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--lattice", type=int, default=LATTICE, help="nodes per side of the synthetic M6-L0-like level")
    ap.add_argument("--fast", action="store_true", help="allow FMA contraction (MGCFD_OPT_EXACT=0)")
    ap.add_argument("--variant", type=int, default=-1, help="MGCFD_OPT_FLUX_VARIANT (-1 automatic, 0 stream k, 1 recompute k, 2/3 edge-once tiles)")
    ap.add_argument("--vcycle", action="store_true",
                    help="also measure wall seconds per 4-level MG V-cycle (off by default so that a rocprofv3 "
                         "summary of the default command holds only the timed workload's launches)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="bounded CPU-baseline sample (0 disables)")
    args = ap.parse_args()

    # multi-process GPU work on this platform needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails otherwise); the image
    # exports it, keep it if a launcher dropped the environment
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 code path on a one-GPU box (not a measurement): every rank on device 0,
    # collectives through gloo.  MGCFD_BENCH_REHEARSAL=1 python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2
    rehearsal = os.environ.get("MGCFD_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MG-CFD HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the communication libraries announce themselves on stdout ("Librccl path : ...", "[Gloo] Rank ..."): keep
        # stdout for the one JSON line by pointing file descriptor 1 at stderr while they initialise
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearsal:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            warm = torch.zeros(1, dtype=torch.float64, device=torch.device("cuda", local_rank))
            dist.all_reduce(warm)                       # the first collective loads and sets up the backend
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    import mgcfd
    mg, levels = build_workload(args.lattice)
    solver = mgcfd.Solver.from_arrays(levels, mg.mesh_variant, device=local_rank)
    # one explicit stream for the solver's kernels AND torch's collectives (the legacy default stream
    # cannot be shared with the library: its own stream does not synchronise with it)
    stream = torch.cuda.Stream(device=local_rank)
    torch.cuda.set_stream(stream)
    solver.set_stream(stream.cuda_stream)
    solver.set_option("exact", 0 if args.fast else 1)
    solver.set_option("flux_variant", args.variant)
    nel, n_int = solver.nel(0), solver.num_internal_edges(0)
    solver.set(0, "variables", perturbed_state(nel, solver.far_field()[:5]))

    sharded = None
    if world > 1:
        from mgcfd.distributed import HipSolverAdapter, ShardedSweep
        sharded = ShardedSweep(HipSolverAdapter(solver, torch.device("cuda", local_rank)), dist)

    def step():
        if world == 1:
            solver.smooth(0, 1)
        else:
            # same sweep, with the global-min time step reduced over all ranks' mesh copies
            sharded.sweep(0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    solver.reset_monitoring()
    if os.environ.get("MGCFD_BENCH_NO_TIMING") != "1":      # (diagnostic: how much the live kernel timing costs)
        solver.set_option("timing", 2)        # hipEvent pairs around the flux launches only
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    flux_avg, flux_launches = solver.flux_kernel_time(0)
    solver.set_option("timing", 0)
    rc, bad = solver.check_for_invalid_variables(0)
    if rc != 0:
        raise SystemExit(f"state became invalid during the bench (code {rc}, cell {bad})")

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # the standalone flux kernel (compute_flux_edge semantics: writes fluxes[]), 50 back-to-back
    # launches between one hipEvent pair on the same stream
    flux_only = solver.bench_flux(0, 50)

    # HBM-side traffic per launch from the committed PMC profile of this same command (bench.py
    # cannot collect hardware counters itself); None when the workload differs from the profiled one
    traffic = {}
    try:
        if args.lattice == LATTICE and not args.fast and args.variant in (-1, 0):   # -1 resolves to 0 at this size
            traffic = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))
    except (OSError, ValueError):
        traffic = {}

    if rank == 0:
        edges_total = 3 * n_int * args.steps * world
        # ALGORITHMIC bytes: compute_flux_edge 40*E + 80*N; time_step 168 B/node.
        # One k_flux_tile<FUSE> launch is a whole RK stage = both loops.
        bytes_flux = 40 * n_int + 80 * nel
        bytes_ts = 168 * nel
        algo_bytes = bytes_flux + bytes_ts
        achieved = algo_bytes / flux_avg / 1e9 if flux_avg > 0 else 0.0
        achieved_flux_only = bytes_flux / flux_only / 1e9 if flux_only > 0 else 0.0
        out = {
            "metric": "Medges/s (compute_flux_edge)",
            "value": round(edges_total / elapsed / 1e6, 3),
            "unit": "Medges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 6),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"M6-L0-like synthetic level ({args.lattice}^3 jittered lattice, permuted ids): "
                                   f"{nel} nodes / {n_int} internal edges per GPU, flux + update sweep, no MG",
                       "step": "copy, compute_step_factor, 3 x (fluxes, time_step), residual (the reference's per-sweep loops; run as 3 fused launches)",
                       "numerics": "fast (FMA contraction)" if args.fast else "exact (bit-identical to the reference)",
                       "parallelism": f"{world} mesh copies, all-reduce(min dt) per sweep" if world > 1 else "1 GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic.get("k_flux_tile_fused_stage", {}).get("bytes"),
                         "kernel": "k_flux_tile<FUSE>: one RK stage per launch = compute_flux_edge + boundary + wall fluxes + time_step",
                         "launches": flux_launches, "avg_kernel_us": round(flux_avg * 1e6, 3),
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "algorithmic_bytes": {"compute_flux_edge (40E+80N)": bytes_flux, "time_step (168N)": bytes_ts},
                         "frac_if_priced_as_flux_only": round(bytes_flux / flux_avg / 1e9 / HBM_PEAK_GBS, 4) if flux_avg > 0 else None,
                         "flux_kernel_alone": {"kernel": "k_flux_tile (writes fluxes[], no time_step)",
                                               "avg_kernel_us": round(flux_only * 1e6, 3), "launches": 50,
                                               "algorithmic_bytes_per_launch": bytes_flux,
                                               "achieved": round(achieved_flux_only, 1),
                                               "traffic": traffic.get("k_flux_tile_flux_only", {}).get("bytes"),
                                               "frac": round(achieved_flux_only / HBM_PEAK_GBS, 4),
                                               "medges_per_s": round(n_int / flux_only / 1e6, 1) if flux_only > 0 else None}},
        }
        if world == 1 and args.vcycle:
            out["vcycle"] = vcycle_wall(args.fast)
        if args.cpu_seconds > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(levels, args.cpu_seconds)
        print(json.dumps(out))
    solver.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
