#!/usr/bin/env python3
"""bench.py — MG-CFD hot path on MI355X: Medges/s of compute_flux_edge + wall seconds per MG V-cycle.

Contract (one JSON line on rank 0):
  python bench.py --gpus N --steps K --warmup W
      N = 1: runs in this process.
      N > 1 and no WORLD_SIZE in the environment: bench.py starts N child ranks itself (a fresh
              `python -m torch.distributed.run --nproc-per-node N ... bench.py <same flags>` child process — never a
              re-exec of a process that has touched the GPU), relays the child's JSON line and exits with its code;
              fewer than N visible GPUs is an error (exit 2), never a silent one-rank run.
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...      (what the driver does for N > 1)
      every rank checks WORLD_SIZE == --gpus and refuses to run otherwise.

Workloads (--workload; `auto` = level0 at N = 1, partitioned at N > 1):
  level0         BASELINE.json configs[1], "Onera-M6 L0 only, flux+update kernels only, no MG", in the synthetic form of
                 SURVEY.md §8d cfg2 (the dataset release is not available): a 67^3 jittered lattice (300,763 nodes /
                 888,822 internal edges), node ids randomly permuted, mesh_name = m6wing, state = far field +-1 % noise.
                 One STEP = one smoothing sweep on that level, the per-level body of the reference's cycle loop
                 (src/euler3d_cpu_double.cpp:383-508): copy, compute_step_factor, 3 x [fluxes, time_step], residual.
                 The same process then measures the second half of BASELINE's metric, wall seconds per 4-level V-cycle
                 (configs[2]; `vcycle` in the JSON line; --no-vcycle skips it, e.g. under rocprofv3).
  partitioned    configs[4]: the level tiled 8x (134^3 lattice = 2.4 M nodes / 7.2 M edges, connected), split over the
                 N ranks by recursive coordinate bisection; every rank sweeps its owned nodes with one fused launch per
                 Runge-Kutta stage, a halo message to every neighbouring rank after each stage and one all-reduce(MIN)
                 of the time step per sweep.  Total work fixed: "scaling": "strong".  The loop runs inside the library
                 (--exchange auto): direct stores into the neighbours' memory over xGMI (HIP IPC) if that form reproduces
                 the torch path bit for bit at start-up AND over the whole run, else RCCL send/receive, else torch.
  copies         the reference's own -m N mesh duplication (src/Base/io_enhanced.cpp:89-201): one level0 mesh per rank,
                 coupled only by the global-min time step (src/Kernels/cfd_loops.cpp:137-150).  "scaling": "weak".
  level-per-gpu  configs[3]: the 4-level hierarchy with level l on rank l % N; a STEP is one V-cycle; restricted
                 variables / coarse residuals move between ranks as whole-array RCCL point-to-point messages.

value = internal edges pushed through compute_flux_edge by all ranks / wall time of the K timed steps (max over ranks).

roofline: the kernel BASELINE's target names, compute_flux_edge (+ boundary + far-field faces) = k_flux_tile writing
fluxes[], priced as SURVEY.md §8d prices it (40 B per edge + 80 B per node) against 8 TB/s, its duration measured
here with hipEvents on the launch stream; `fused_stage` beside it is the launch the sweeps actually run (one whole
Runge-Kutta stage = compute_flux_edge + time_step, 168 B per node more), timed by ONE hipEvent pair around the K
timed steps (3 such launches per step, back to back; nothing else is enqueued inside the region).  `traffic` is
not measurable from inside a process: it is the figure of the committed rocprofv3 --pmc profile of this command,
labelled with the file and build it came from.
cpu_baseline: the reference's own compute_flux_edge (oracle/_ref, built from the reference sources) — or the C oracle
port when that build is absent — timed on one host core on the same mesh for a bounded number of passes; beside it the
all-cores figure (the reference's race-free way to use threads: one private mesh copy per thread, its -m).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd"))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
HBM_COPY_GBS = 6290.0        # ... and what a device-to-device copy measures there ("report both")
LATTICE = 67                 # 67^3 = 300,763 nodes, 888,822 internal edges
LATTICE_8X = 134             # the level tiled 8x (connected): 2,406,104 nodes / 7,164,444 internal edges
HIERARCHY = (67, 55, 48, 43) # 300,763 / 166,375 / 110,592 / 79,507 nodes
HIERARCHY_8X = (134, 110, 96, 86)   # the same hierarchy tiled 8x (connected): 2,406,104 / 1,331,000 / 884,736 / 636,056 nodes
TRAFFIC_PROFILE = os.path.join("profiles", "r4_traffic.json")
ROOFLINE_LAUNCHES = 1000     # back-to-back launches of the standalone flux kernel / of its data-movement probe per measurement
PREHEAT_SWEEPS = 1000           # untimed sweeps in front of a short timed region (see main)


def build_workload(lattice: int, seed: int = 0, mesh: str = "lattice"):
    """mesh = "lattice": the jittered n^3 lattice (every interior node has six neighbours); "mixed": the same points with the
    connectivity of three element types — hexahedral core, prism layers on a wall, a tetrahedral far field: internal degrees
    3 ... 14, 18 % of the nodes not 6 (mgcfd.meshgen.make_mixed_level; 67^3: 300,763 nodes / 1,004,901 internal edges)."""
    from mgcfd import meshgen, generated_to_levels
    if mesh == "mixed":
        mg = meshgen.make_mixed_multigrid((lattice,), "m6wing", seed=seed, jitter=0.2, area_noise=0.02, volume_noise=0.02, permute=True)
    else:
        mg = meshgen.make_multigrid((lattice,), "m6wing", seed=seed, jitter=0.2, area_noise=0.02,
                                    volume_noise=0.02, permute=True)
    return mg, generated_to_levels(mg)


def mixed_mesh_roofline(make_solver, fast: bool):
    """The headline kernel on a level of NON-UNIFORM degree beside the lattice's figure: compute_flux_edge (+ boundary + far-field
    faces) on the mixed-element level, priced as the lattice's (40 B per internal edge + 80 B per node against 8 TB/s)."""
    mg, levels = build_workload(LATTICE, mesh="mixed")
    s = make_solver(levels, mg.mesh_variant)
    nel, n_int = s.nel(0), s.num_internal_edges(0)
    s.set(0, "variables", perturbed_state(nel, s.far_field()[:5]))
    til = s.tiling(0)
    s.bench_flux(0, ROOFLINE_LAUNCHES)
    t = s.bench_flux(0, ROOFLINE_LAUNCHES)
    probe = s.bench_indirect_rw(0, ROOFLINE_LAUNCHES)
    s.zero_fluxes(0)
    s.smooth(0, 200); s.synchronize()
    t0 = time.perf_counter(); s.smooth(0, 1000); s.synchronize(); sweep = (time.perf_counter() - t0) / 1000
    algo = 40 * n_int + 80 * nel
    out = {"mesh": f"mixed-element level on the {LATTICE}^3 points (hexahedral core, prism layers on a wall, tetrahedral far field; permuted ids): {nel} nodes / "
                   f"{n_int} internal edges, internal degrees 3 ... 14",
           "kernel": "compute_flux_edge + boundary + far-field faces in one launch (the same instantiation as on the lattice, or its long-row form)",
           "avg_kernel_us": round(t * 1e6, 3), "launches": ROOFLINE_LAUNCHES, "algorithmic_bytes_per_launch": algo,
           "achieved": round(algo / t / 1e9, 1), "frac": round(algo / t / 1e9 / HBM_PEAK_GBS, 4), "medges_per_s": round(n_int / t / 1e6, 1),
           "empirical_ceiling_us": round(probe * 1e6, 3), "sweep_us": round(sweep * 1e6, 2), "sweep_medges_per_s": round(3 * n_int / sweep / 1e6, 1),
           "tiling": {"tiles": til["tiles"], "halo_max": til["halo_max"], "halo_mean": round(til["halo_nodes"] / max(til["tiles"], 1), 1),
                      "ell_padding": round(til["padding_entries"] / max(til["row_entries"], 1), 4), "long_row_list_entries": til["list_entries"],
                      "overflow_refs": til["overflow_refs"]},
           "numerics": "fast (FMA contraction)" if fast else "exact (bit-identical to the reference: tests/test_gpu_configs.py::test_mixed_level_full_size_sweep)"}
    if not fast and s.has_order_free(0):
        s.set_option("exact", 0); s.set_option("flux_variant", 65)
        s.bench_flux(0, ROOFLINE_LAUNCHES)
        tf = s.bench_flux(0, ROOFLINE_LAUNCHES)
        out["order_free"] = {"avg_kernel_us": round(tf * 1e6, 3), "frac": round(algo / tf / 1e9 / HBM_PEAK_GBS, 4),
                             "note": "k_flux_free on this level (slices of more than five half rows per lane walk the rest in a loop); tests/test_gpu_order_free.py"}
    s.close()
    return out


def build_hierarchy(sizes=HIERARCHY):
    from mgcfd import meshgen, generated_to_levels
    mg = meshgen.make_multigrid(sizes, "m6wing", seed=0, jitter=0.2, area_noise=0.02, volume_noise=0.02)
    return mg, generated_to_levels(mg)


def injected_failure(where: str, rank: int = 0):
    """MGCFD_BENCH_FAIL = a comma list of places where THIS run pretends something breaks on ONE rank (the last), so that the
    form ladder's every way out can be rehearsed (tests/test_gpu_configs.py): `attach` the library's rank set-up raises, `phase`
    the library's start-up sweep raises, `vcycle-setup` / `vcycle-cycles` the same in the V-cycle leg; `ipc-start` / `ipc-end`
    (also MGCFD_BENCH_FAIL_IPC=start|end) make the IPC form's checks come out negative."""
    want = [w.strip() for w in os.environ.get("MGCFD_BENCH_FAIL", "").split(",") if w.strip()]
    if os.environ.get("MGCFD_BENCH_FAIL_IPC") in ("start", "end"):
        want.append("ipc-" + os.environ["MGCFD_BENCH_FAIL_IPC"])
    if where in ("ipc-start", "ipc-end"):
        return where in want
    if where in want and rank == int(os.environ.get("WORLD_SIZE", "1")) - 1:
        raise RuntimeError(f"injected failure ({where}) on rank {rank}")
    return False


def injected(where: str) -> bool:
    """`where` is named in MGCFD_BENCH_FAIL (the legs one process runs alone: no rank to pick)."""
    return where in [w.strip() for w in os.environ.get("MGCFD_BENCH_FAIL", "").split(",")]


def perturbed_state(nel, ff_var, seed=1234, amplitude=0.01):
    rng = np.random.default_rng(seed)
    base = np.tile(np.asarray(ff_var, dtype=np.float64), (nel, 1))
    return base * (1.0 + amplitude * rng.uniform(-1.0, 1.0, base.shape))


def cpu_model():
    """CPU model string and core counts of the host, as the reference records them (src/Base/common.h:114-143)."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count() or 0, len(os.sched_getaffinity(0))


def cpu_baseline(levels, sample_seconds: float):
    """Time compute_flux_edge on ONE host core for a bounded number of passes, then on every core this process may use."""
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
    have_ref = O.have_reference()
    if have_ref:
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
    model, cores_total, cores_usable = cpu_model()
    out = {"value": round(n_int * passes / dt / 1e6, 3), "unit": "Medges/s", "cores": 1, "kind": kind,
           "sample": f"{passes} passes of compute_flux_edge over the same {n_int}-edge level ({dt:.1f} s), {how}",
           "cpu_model": model, "host_cores": cores_total, "cores_usable_by_this_process": cores_usable}
    # ... and on all the host cores this process may use, the reference's own race-free way to use threads: one
    # private copy of the mesh state per thread (its -m duplication, src/Base/io_enhanced.cpp:89-201), here as one
    # thread per copy calling the reference's compute_flux_edge on its own arrays (the reference's loop counters are
    # process-wide statistics nobody reads here; its arithmetic touches only the arrays passed in).
    try:
        import threading
        # 16 threads unless MGCFD_BENCH_CPU_THREADS says otherwise: a one-GPU box's CPU share is 16 cores whatever the host has and
        # whatever the affinity mask says (64 threads there: 186 Medges/s in 42 s against 377 in 11 s with 16, profiles/r4_*)
        cores = max(1, min(cores_usable, int(os.environ.get("MGCFD_BENCH_CPU_THREADS", "16"))))
        per = max(2, int(sample_seconds / 4.0 / max(one, 1e-6)))
        state = [(q.copy(), np.zeros_like(q)) for _ in range(cores)]
        if have_ref and os.environ.get("MGCFD_BENCH_ALLCORES_PORT") != "1":
            call, all_kind, all_how = (lambda qk, fk: ref.ref_compute_flux_edge(0, n_int, O.ptr(edges), O.ptr(qk), O.ptr(fk))), "reference", how
        else:
            call, all_kind, all_how = (lambda qk, fk: lib.ora_compute_flux_edge(0, n_int, O.ptr(edges), O.ptr(qk), O.ptr(fk))), "port", \
                "oracle/mgcfd_oracle.c ora_compute_flux_edge, gcc -O3 -fno-fast-math -march=native"

        def work(k):
            qk, fk = state[k]
            for _ in range(per):
                call(qk, fk)

        threads = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
        t0 = time.perf_counter()
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        dt_all = time.perf_counter() - t0
        out["multi_thread"] = {"value": round(n_int * per * cores / dt_all / 1e6, 3), "unit": "Medges/s", "threads": cores, "cores": cores, "kind": all_kind,
                               "cores_usable_by_this_process": cores_usable, "host_cores": cores_total,
                               "threads_note": "16 by default = the CPU share of a one-GPU box (MGCFD_BENCH_CPU_THREADS overrides); not the host's core count",
                               "sample": f"{cores} threads x {per} passes, one private copy of the level's state per thread ({dt_all:.1f} s), {all_how}"}
    except Exception as e:                                   # the single-core figure stands on its own
        out["multi_thread"] = {"error": str(e)}
    return out


def vcycle_wall(fast: bool, cycles: int = 25, device: int = 0):
    """Second half of BASELINE.json's metric: wall seconds per MG V-cycle on the 4-level M6-like hierarchy
    (67^3/55^3/48^3/43^3 lattices = 300,763/166,375/110,592/79,507 nodes, nearest-node maps,
    mesh_name = m6wing), 25 cycles as the reference's default (src/Base/config.cpp:63), best of 5 — the reference's
    "Total" / cycles (src/Monitoring/timer.cpp:106-195) with everything resident."""
    import mgcfd
    mg, levels = build_hierarchy()
    s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant, device=device)
    s.set_option("exact", 0 if fast else 1)
    s.run_cycles(100)                       # ~30 ms of load: the clocks of an idle MI355X are up only after that (tools/exp/first_steps.py)
    best = float("inf")
    for _ in range(5):
        s.reset_monitoring()
        t0 = time.perf_counter()
        rms = s.run_cycles(cycles)
        best = min(best, time.perf_counter() - t0)
    edge_iters = sum(s.loop_iters(l)["flux"] for l in range(s.num_levels)) // cycles
    out = {"workload": f"4-level M6-like synthetic hierarchy {[l.nel for l in mg.levels]} nodes, {cycles} cycles, best of 5 after 100 untimed cycles",
           "wall_s_per_cycle": round(best / cycles, 9), "flux_edge_iterations_per_cycle": edge_iters,
           "medges_per_s_whole_cycle": round(edge_iters * cycles / best / 1e6, 1), "rms_last": float(rms[-1])}
    if not fast:
        # beside it, never as it: the same cycles with MGCFD_OPT_EXACT = 0 (FMA contraction; <= 1e-12 relative per sweep of the
        # reference, tests/test_gpu_parity.py REL_FAST — north_star allows 1e-10)
        s.set_option("exact", 0)
        s.run_cycles(25)
        best_c = float("inf")
        for _ in range(5):
            t0 = time.perf_counter()
            rms_c = s.run_cycles(cycles)
            best_c = min(best_c, time.perf_counter() - t0)
        out["fma_contracted"] = {"wall_s_per_cycle": round(best_c / cycles, 9), "rms_last": float(rms_c[-1])}
        s.set_option("exact", 1)
    s.close()
    return out


def vcycle_partitioned(args, dist, world, rank, dev, stream, rehearsal, sizes, cycles=25, keep=None):
    """The V-cycle half of BASELINE's metric on N GPUs: the 4-level hierarchy (tiled 8x by default: per-rank work as on one GPU)
    with EVERY level partitioned over the ranks (mgcfd.partition.partition_hierarchy, level 0 by recursive coordinate bisection),
    the whole cycle swept inside the library (mgcfd_rank_cycles: the partitioned sweeps of `partitioned` on every level, halo
    messages of the coarse `variables` after mg_restrict, of the coarse `residuals` before the prolongation and of the fine
    `variables` after it, over RCCL send / receive).  Before it is timed it must reproduce, on every rank and bit for bit, two
    cycles of the torch.distributed orchestration (mgcfd.distributed.PartitionedCycle) from the same state; otherwise — and in the
    one-GPU rehearsal, where RCCL cannot form a communicator — the torch path is what is timed, and the line says so."""
    import torch
    import mgcfd
    from mgcfd.distributed import HipSolverAdapter, PartitionedCycle
    from mgcfd.partition import partition_hierarchy, rcb_partition
    t0 = time.perf_counter()
    mg, levels = build_hierarchy(sizes)
    Hs = partition_hierarchy(levels, rcb_partition(np.asarray(levels[0]["coords"]), world))
    H = Hs[rank]
    lv, owned, keys = H.solver_args()
    nels, n_ints = [int(L["nel"]) for L in levels], [int(L["n_internal"]) for L in levels]
    if keep is not None:                                    # (rank 0's in-process group leg splits the same hierarchy the same way)
        keep.append((mg, Hs, nels, n_ints))
    del levels, Hs
    s = mgcfd.Solver.from_arrays(lv, mg.mesh_variant, device=dev.index, n_owned=owned, order_keys=keys)
    s.set_stream(stream.cuda_stream)
    s.set_option("exact", 0 if args.fast else 1)
    setup_s = time.perf_counter() - t0
    nlev = len(lv)
    ff = s.far_field()[:5]
    cyc = PartitionedCycle(HipSolverAdapter(s, dev), H, dist, make_buffer=lambda n: torch.empty(max(n, 1), dtype=torch.float64, device=dev), fused=True)
    notes = []
    use_library = False

    def reset(exchange):
        torch.cuda.synchronize(); dist.barrier()
        for l in range(nlev):
            s.set(l, "variables", np.tile(ff, (int(lv[l]["nel"]), 1)))
        for l in range(nlev):
            exchange(l)

    def agree(ok):
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item()) == 1.0

    torch_exchange = lambda l: cyc.exchange(l, "variables")
    # Every step only ONE rank can see failing is followed by an agreement (a torch all-reduce) BEFORE any rank makes the next call
    # that involves the others: a rank that cannot go on then keeps every rank out of the library's RCCL loop instead of
    # leaving them in a send / receive nobody answers (round 3's advisor finding: the `vcycle-cycles` injection hung the run).
    # The rehearsal on one GPU (RCCL cannot form a communicator with both ranks on one device) goes through the same
    # agreements and takes the same ways out.
    err = None
    uid = [None]
    if not rehearsal:
        try:
            uid = [mgcfd.rccl_unique_id() if rank == 0 else None]
        except Exception as e:
            err = e
        dist.broadcast_object_list(uid, src=0)
    try:
        if err:
            raise err
        injected_failure("vcycle-setup", rank)
        if not rehearsal:
            s.rank_attach_rccl(rank, world, uid[0])
            for l in range(nlev):
                s.rank_set_halo(l, H.levels[l])
    except Exception as e:
        err = e
    if not agree(err is None):
        notes.append(f"the library's cycle loop not used: {err or 'its set-up failed on another rank'}")
    else:
        # two cycles of each form from the far field: level 0's owned nodes and ghosts must be the same bits
        reset(torch_exchange)
        for _ in range(2):
            cyc.cycle()
        torch.cuda.synchronize()
        want = [s.get(l, "variables") for l in range(nlev)]
        err = None
        try:
            injected_failure("vcycle-cycles", rank)            # (the pre-flight point: whatever a rank finds wrong before the loop)
        except Exception as e:
            err = e
        if not agree(err is None):
            notes.append(f"mgcfd_rank_cycles not used: {err or 'a rank could not start it'}")
        else:
            try:
                if rehearsal:
                    raise RuntimeError("rehearsal on one GPU: RCCL cannot form a communicator there, the torch.distributed orchestration is timed")
                reset(lambda l: s.rank_exchange(l))
                s.rank_cycles(2, rms=False)
                same = all(np.array_equal(s.get(l, "variables").view(np.int64), want[l].view(np.int64)) for l in range(nlev))
            except Exception as e:
                err, same = e, False
            use_library = agree(same)
            if not use_library:
                notes.append(f"mgcfd_rank_cycles not used: {err or 'its two cycles differ from the torch orchestration on some rank'}")
    step = (lambda n: s.rank_cycles(n, rms=False)) if use_library else (lambda n: [cyc.cycle() for _ in range(n)])
    reset((lambda l: s.rank_exchange(l)) if use_library else torch_exchange)
    step(cycles)                                            # (untimed: clocks up, every buffer and graph in place)
    best = float("inf")
    for _ in range(3):
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        step(cycles)
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        best = min(best, float(t.item()))
    rc, bad = s.check_for_invalid_variables(0)
    edge_iters = 3 * (n_ints[0] + n_ints[-1] + 2 * sum(n_ints[1:-1]))
    out = {"workload": f"4-level M6-like synthetic hierarchy {nels} nodes, every level partitioned over {world} ranks (level 0 by recursive coordinate "
                       f"bisection, a coarse node with its first child), {cycles} cycles, best of 3 after {cycles} untimed",
           "wall_s_per_cycle": round(best / cycles, 9), "flux_edge_iterations_per_cycle": edge_iters,
           "medges_per_s_whole_cycle": round(edge_iters * cycles / best / 1e6, 1),
           "form": ("libmgcfd_hip (mgcfd_rank_cycles): the cycle's state machine and every exchange inside the library over RCCL; two cycles checked against the "
                    "torch.distributed orchestration bit for bit on every rank before timing") if use_library else
                   "torch.distributed orchestration from Python (mgcfd.distributed.PartitionedCycle): fused stage launches, pack / batch_isend_irecv / unpack per peer",
           "owned_nodes_rank0": [int(v) for v in owned], "local_nodes_rank0": [int(L["nel"]) for L in lv],
           "state_valid": rc == 0, "setup_s_rank0": round(setup_s, 1), "notes": notes}
    try:
        if use_library:
            s.rank_detach()
    except Exception:
        pass
    s.close()
    return out


class leg_guard:
    """A leg that only adds to the line (the V-cycle leg of an N > 1 run, the in-process group leg) must not take the measured
    sweeps down with it: if it has not ended after `seconds`, rank 0 prints the line it has — the leg's record saying what
    happened — and every rank leaves through os._exit (a rank stuck inside a collective cannot be interrupted any other way).
    Nothing is re-executed; a process that has touched the GPU is never replaced."""

    def __init__(self, seconds, rank, line, key, what):
        import threading
        self.timer = threading.Timer(seconds, self.fire)
        self.timer.daemon = True
        self.rank, self.line, self.key, self.what, self.seconds = rank, line, key, what, seconds

    def fire(self):
        sys.stderr.write(f"bench.py: rank {self.rank}: {self.what} did not end within {self.seconds:.0f} s; leaving with what was measured\n")
        sys.stderr.flush()
        line = self.line() if callable(self.line) else self.line            # (a callable: the line as it stands when the guard fires)
        if self.rank == 0 and line is not None:
            line[self.key] = {"error": f"{self.what} did not end within {self.seconds:.0f} s on some rank; the line's other figures were measured before it"}
            sys.stdout.write(json.dumps(line) + "\n")
            sys.stdout.flush()
        os._exit(0 if line is not None or self.rank != 0 else 1)

    def cancel(self):
        self.timer.cancel()

    def __enter__(self):
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()


def two_launches_in_flight(solver, levels, mesh_variant, device, fast: bool):
    """What the flux kernel does when the chip is kept full: batches of the standalone launch from TWO solvers on the same level at
    once, each on its own stream from its own host thread — independent launches, which the hardware overlaps (one launch's tail and
    boundary under the other's rows) — as wall time per launch over both batches.  Never the line's `roofline.frac` (that is ONE
    launch after the other, as a sweep needs them); it separates what the kernel's code reaches from what the shape of a 1.15-1.5
    round launch costs (tools/exp/two_streams.py, profiles/r4_dataflow.txt).  Leaves the solver's options as it found them."""
    import threading
    import mgcfd
    out = {"what": "two independent batches of the standalone compute_flux_edge launch in flight at once (two solvers on the level, two streams, "
                   "two host threads): wall time over both batches / launches of both; frac = 40 E + 80 N per launch against 8 TB/s.  What the "
                   "kernel reaches when the chip is kept full — mesh copies, several meshes, ranks sharing a device — not a single sweep's figure"}
    other = None
    try:
        other = mgcfd.Solver.from_arrays(levels, mesh_variant, device=device)            # (its own stream)
        other.set(0, "variables", solver.get(0, "variables"))
        pair = (solver, other)
        modes = [("order_free", 0, 65)] if fast else [("bit_identical", 1, 1), ("order_free", 0, 65)]
        for name, ex, var in modes:
            if var == 65 and not solver.has_order_free(0):
                continue
            for s in pair:
                s.set_option("exact", ex); s.set_option("flux_variant", var); s.bench_flux(0, ROOFLINE_LAUNCHES // 2)
            best = None
            failed = []

            def batch(s):
                try:
                    s.bench_flux(0, ROOFLINE_LAUNCHES)
                except Exception as e:                           # (a thread's exception would otherwise only be printed)
                    failed.append(e)
            for _ in range(3):
                th = [threading.Thread(target=batch, args=(s,)) for s in pair]
                t0 = time.perf_counter()
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                if failed:
                    raise failed[0]
                dt = (time.perf_counter() - t0) / (2 * ROOFLINE_LAUNCHES)
                best = dt if best is None else min(best, dt)
            out[name] = {"us_per_launch": round(best * 1e6, 3), "launches": 2 * ROOFLINE_LAUNCHES}
        # ... and whole sweeps (three fused stage launches each) of the mode the line is quoted in: one solver alone, both at once
        n_sw = 500
        for s in pair:
            s.set_option("exact", 0 if fast else 1); s.set_option("flux_variant", -1)
            s.zero_fluxes(0); s.smooth(0, 100); s.synchronize()
        t0 = time.perf_counter(); solver.smooth(0, n_sw); solver.synchronize(); alone = (time.perf_counter() - t0) / n_sw

        def sweeps(s):
            try:
                s.smooth(0, n_sw); s.synchronize()
            except Exception as e:
                failed.append(e)
        failed = []
        best = None
        for _ in range(2):
            th = [threading.Thread(target=sweeps, args=(s,)) for s in pair]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            if failed:
                raise failed[0]
            dt = (time.perf_counter() - t0) / (2 * n_sw)
            best = dt if best is None else min(best, dt)
        out["sweeps"] = {"us_per_sweep_one_solver": round(alone * 1e6, 3), "us_per_sweep_two_at_once": round(best * 1e6, 3),
                         "gain": round(alone / best, 3), "sweeps": 2 * n_sw,
                         "what": "two solvers sweeping the level at once (wall over both / sweeps of both) against one alone: what independent "
                                 "sweeps — mesh copies, several meshes — gain from filling each other's launch boundaries and tails"}
    except Exception as e:
        out["error"] = f"{type(e).__name__}: {e}"
    finally:
        if other is not None:
            other.close()
        solver.set_option("exact", 0 if fast else 1)
        solver.set_option("flux_variant", -1)
    return out


def in_process_group_leg(args, world, lattice, level_built, sizes, hierarchy_built, share_device: bool, steps: int, warmup: int, cycles: int = 25):
    """The form `euler3d_gpu_double --gpus N` runs, timed from ONE process: a solver per device as the ranks of an
    in-process group (mgcfd_group_*: a host thread per rank issues that rank's launches, a stage's message is one launch that
    stores straight into the neighbours' ghost slots over xGMI peer access, the time-step minimum read from the peers' scalars —
    no collective library, no second process).  Sweeps on the 8x level in N parts (mgcfd_group_sweeps) and V-cycles on the
    partitioned hierarchy (mgcfd_group_cycles); before a figure counts, the owned nodes of every rank after the timed sweeps
    must equal the UNPARTITIONED level swept on device 0 bit for bit (the check the IPC forms get).  `share_device`: every rank
    on device 0 (the one-GPU rehearsal: a functional run, not a measurement)."""
    import mgcfd
    from mgcfd.partition import partition_hierarchy, partition_level, rcb_partition
    out = {"form": "libmgcfd_hip in-process group (mgcfd_group_sweeps / mgcfd_group_cycles): one process, a host thread per rank, a stage's message = one "
                   "launch storing into the neighbours' ghost slots (peer access over xGMI), min dt read from the peers' scalars"
                   + ("; REHEARSAL: every rank on device 0" if share_device else ""), "ranks": world}
    if injected("group-setup"):
        raise RuntimeError("injected failure (group-setup)")
    device_of = (lambda r: 0) if share_device else (lambda r: r)
    fast = 0 if args.fast else 1
    # ---- sweeps on the level in N parts ----
    mg, levels = level_built if level_built is not None else build_workload(lattice)
    L = levels[0]
    nel, n_int = int(L["nel"]), int(L["n_internal"])
    parts = partition_level(L, rcb_partition(np.asarray(L["coords"]), world))
    whole = mgcfd.Solver.from_arrays([L], mg.mesh_variant, device=0)
    whole.set_option("exact", fast)
    q0 = perturbed_state(nel, whole.far_field()[:5])
    whole.set(0, "variables", q0)
    whole.smooth(0, warmup + steps)
    want = whole.get(0, "variables")
    whole.close()
    solvers = []
    for P in parts:
        sv = mgcfd.Solver.from_arrays([P.level], mg.mesh_variant, device=device_of(P.rank), n_owned=[P.n_owned])
        sv.set_option("exact", fast)
        sv.set(0, "variables", q0[P.global_ids])
        solvers.append(sv)
    g = mgcfd.Group(solvers)
    for P, sv in zip(parts, solvers):
        sv.rank_set_halo(0, P)
    g.exchange(0)
    g.sweeps(0, warmup); g.synchronize()
    t0 = time.perf_counter()
    g.sweeps(0, steps); g.synchronize()
    dt = time.perf_counter() - t0
    bad = sum(int(np.count_nonzero(np.any(sv.get(0, "variables")[:P.n_owned].view(np.int64) != want[P.global_ids[:P.n_owned]].view(np.int64), axis=1)))
              for P, sv in zip(parts, solvers))
    if injected("group-end"):
        bad += 1
    out["sweeps"] = {"workload": f"{lattice}^3 level ({nel} nodes / {n_int} internal edges) in {world} parts, {steps} sweeps after {warmup}",
                     "ms_per_step": round(dt / steps * 1e3, 6), "value": round(3 * n_int * steps / dt / 1e6, 3), "unit": "Medges/s",
                     "nodes_differing_from_the_unpartitioned_level": bad,
                     "counts": bad == 0}
    # a longer batch beside the contract's K steps: the host threads start once per call
    if bad == 0 and steps < 200:
        g.sweeps(0, 200); g.synchronize()
        t0 = time.perf_counter(); g.sweeps(0, 200); g.synchronize()
        out["sweeps"]["ms_per_step_200_sweeps"] = round((time.perf_counter() - t0) / 200 * 1e3, 6)
    g.close()
    for sv in solvers:
        sv.close()
    del levels, parts, solvers, want
    # ---- V-cycles on the partitioned hierarchy ----
    if sizes is not None:
        if hierarchy_built is not None:
            mg, Hs, nels, n_ints = hierarchy_built
        else:
            mg, levels = build_hierarchy(sizes)
            nels, n_ints = [int(l["nel"]) for l in levels], [int(l["n_internal"]) for l in levels]
            Hs = partition_hierarchy(levels, rcb_partition(np.asarray(levels[0]["coords"]), world))
            del levels
        ff = None
        solvers = []
        for H in Hs:
            lv, owned, keys = H.solver_args()
            sv = mgcfd.Solver.from_arrays(lv, mg.mesh_variant, device=device_of(H.rank), n_owned=owned, order_keys=keys)
            sv.set_option("exact", fast)
            ff = sv.far_field()[:5]
            for l in range(len(lv)):
                sv.set(l, "variables", np.tile(ff, (int(lv[l]["nel"]), 1)))
            solvers.append(sv)
        g = mgcfd.Group(solvers)
        for H, sv in zip(Hs, solvers):
            for l in range(len(sizes)):
                sv.rank_set_halo(l, H.levels[l])
        for l in range(len(sizes)):
            g.exchange(l)
        g.cycles(cycles, rms=False); g.synchronize()
        best = float("inf")
        for _ in range(3):
            t0 = time.perf_counter()
            g.cycles(cycles, rms=False); g.synchronize()
            best = min(best, time.perf_counter() - t0)
        edge_iters = 3 * (n_ints[0] + n_ints[-1] + 2 * sum(n_ints[1:-1]))
        valid = all(sv.check_for_invalid_variables(0)[0] == 0 for sv in solvers)
        out["vcycle"] = {"workload": f"4-level hierarchy {nels} nodes, every level in {world} parts, {cycles} cycles, best of 3 after {cycles} untimed",
                         "wall_s_per_cycle": round(best / cycles, 9), "medges_per_s_whole_cycle": round(edge_iters * cycles / best / 1e6, 1),
                         "state_valid": valid,
                         "checked_by": "tests/test_gpu_partitioned_cycles.py (bit-identical to mgcfd_run_cycles on the whole hierarchy in 2-8 parts on one GPU)"}
        g.close()
        for sv in solvers:
            sv.close()
    return out


# --------------------------------------------------------------------------------------------------------------
# process plumbing
# --------------------------------------------------------------------------------------------------------------
def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args) -> int:
    """--gpus N > 1 from a plain invocation: start the N ranks as a fresh child process tree and relay its result."""
    rehearsal = os.environ.get("MGCFD_BENCH_REHEARSAL") == "1"
    if not rehearsal and not args.plumbing_only:
        import torch                      # (device_count does not initialise the GPU on this image)
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible; refusing to run fewer ranks than asked "
                  f"(MGCFD_BENCH_REHEARSAL=1 runs every rank on device 0 over gloo as a functional rehearsal)", file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line:
        print(line)
    elif rc == 0:
        print("bench.py: the child ranks printed no result line", file=sys.stderr)
        rc = 1
    return rc


class quiet_stdout:
    """The communication libraries announce themselves on stdout ("Librccl path : ...", "[Gloo] Rank ..."): keep
    stdout for the one JSON line by pointing file descriptor 1 at stderr while they initialise."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)        # 0.13 s of GPU time: past the clock ramp of the first ms
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="auto", choices=["auto", "level0", "partitioned", "partitioned-mg", "copies", "level-per-gpu"])
    ap.add_argument("--vcycle-hierarchy", default="8x", choices=["8x", "base", "tiny"],
                    help="N > 1: the hierarchy of the V-cycle leg — the 4-level hierarchy tiled 8x (per-rank work as on one GPU), the "
                         "one-GPU hierarchy itself (strong scaling of a 0.3 ms cycle), or a (20, 12, 8, 6)^3 one for functional rehearsals")
    ap.add_argument("--lattice", type=int, default=0, help="nodes per side of the synthetic level (default 67; 134 for `partitioned`)")
    ap.add_argument("--mesh", default="lattice", choices=["lattice", "mixed"],
                    help="level0 workload: the jittered lattice (default; every interior node six neighbours) or the mixed-element level on the same points "
                         "(hexahedral core, prism layers, tetrahedral far field: internal degrees 3 ... 14); the default line reports the mixed level's "
                         "flux-kernel figure beside the lattice's in roofline.mixed_mesh")
    ap.add_argument("--fast", action="store_true", help="the fast mode (MGCFD_OPT_EXACT=0): FMA contraction and order-free flux accumulation — within 1e-12 "
                                                        "relative of the reference per sweep, not reproducible bit for bit from run to run")
    ap.add_argument("--variant", type=int, default=-1, help="MGCFD_OPT_FLUX_VARIANT (see include/mgcfd.h)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "library", "ipc", "torch"],
                    help="partitioned workload: who runs the sweep loop and the halo exchange — the library with direct stores into the neighbours' memory "
                         "through HIP IPC (`ipc`), the library over RCCL send/receive (`library`), or torch.distributed from Python (`torch`).  `auto` "
                         "(default) tries them in that order: each must reproduce the torch path's sweep bit for bit at start-up, and the IPC form's "
                         "figure counts only if its final state equals the torch path's after the same sweeps — it has been rehearsed on one GPU only")
    ap.add_argument("--rank-graphs", action="store_true",
                    help="partitioned workload, library exchange: replay every rank's sweep from a captured hipGraph (MGCFD_OPT_GRAPH)")
    ap.add_argument("--no-two-in-flight", action="store_true",
                    help="skip roofline.two_launches_in_flight (two solvers' batches overlapped on purpose: under rocprofv3 --stats their launches "
                         "would raise the per-kernel means the summary is read for)")
    ap.add_argument("--no-rank-graphs", action="store_true", help="N > 1: skip the leg that times the RCCL form's sweeps replayed from hipGraphs (`rank_graphs` in the line)")
    ap.add_argument("--no-group", action="store_true", help="N > 1: skip the in-process group leg (one process sweeping all N devices: `in_process_group` in the line)")
    ap.add_argument("--no-vcycle", action="store_true", help="skip the V-cycle leg (a rocprofv3 summary of the command then holds only the timed workload's launches)")
    ap.add_argument("--vcycle", action="store_true", help="(default now; kept so older command lines still parse)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="bounded CPU-baseline sample (0 disables)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="start the ranks, rendezvous, build the host-side partition and print the line without touching a GPU "
                         "(value null): what the CPU-side test of the N > 1 launcher runs")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    # multi-process GPU work on this platform needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails otherwise); the image
    # exports it, keep it if a launcher dropped the environment
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` or with "
                  f"--nproc-per-node equal to --gpus", file=sys.stderr)
        sys.exit(2)
    workload = args.workload
    if workload == "auto":
        workload = "level0" if world == 1 else "partitioned"
    if workload == "copies" and world == 1:
        workload = "level0"
    if workload == "partitioned-mg":
        # (the sweep leg is `partitioned`'s; the V-cycle leg on the partitioned hierarchy follows it for every N > 1 anyway)
        workload = "partitioned" if world > 1 else "level0"
    lattice = args.lattice or (LATTICE_8X if workload == "partitioned" else LATTICE)
    # rehearsal of the N > 1 code path on a one-GPU box (not a measurement): every rank on device 0, collectives over gloo
    rehearsal = os.environ.get("MGCFD_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0

    import torch
    dist = None
    if world > 1:
        import faulthandler
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # several ranks: a rank stuck in a collective or in an exchange that never completes says where and ends (a whole run
        # takes two to three minutes; nothing multi-GPU has run on hardware yet)
        faulthandler.dump_traceback_later(int(os.environ.get("MGCFD_BENCH_WATCHDOG_S", "900")), exit=True)
    if args.plumbing_only:
        return plumbing_only(args, dist, world, rank, workload, lattice)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MG-CFD HIP path has no CPU fallback")
    if not rehearsal and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks but {torch.cuda.device_count()} visible GPU(s)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        with quiet_stdout():
            if rehearsal:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            warm = torch.zeros(1, dtype=torch.float64, device=dev)
            dist.all_reduce(warm)                       # the first collective loads and sets up the backend
            torch.cuda.synchronize()
    backend = None if world == 1 else ("gloo (rehearsal: every rank on device 0)" if rehearsal else "nccl (RCCL)")

    import mgcfd
    from mgcfd.distributed import HipSolverAdapter, LevelPerRankCycle, PartitionedSweep, ShardedSweep
    # one explicit stream for the solver's kernels AND torch's collectives (the legacy default stream cannot be
    # shared with the library: its own stream does not synchronise with it)
    stream = torch.cuda.Stream(device=local_rank)
    torch.cuda.set_stream(stream)

    def make_solver(levels, variant, **kw):
        s = mgcfd.Solver.from_arrays(levels, variant, device=local_rank, **kw)
        s.set_stream(stream.cuda_stream)
        assert torch.cuda.current_stream().cuda_stream == stream.cuda_stream
        s.set_option("exact", 0 if args.fast else 1)
        s.set_option("flux_variant", args.variant)
        return s

    config = {"numerics": "fast (FMA contraction)" if args.fast else "exact (bit-identical to the reference)"}
    extra = {}
    scaling = "weak"
    edges_per_step_all_ranks = 0
    timed_level = 0
    levels = None

    if workload in ("level0", "copies"):
        mg, levels = build_workload(lattice, mesh=args.mesh)
        solver = make_solver(levels, mg.mesh_variant)
        nel, n_int = solver.nel(0), solver.num_internal_edges(0)
        solver.set(0, "variables", perturbed_state(nel, solver.far_field()[:5]))
        sharded = ShardedSweep(HipSolverAdapter(solver, dev), dist) if world > 1 else None
        step = (lambda: solver.smooth(0, 1)) if world == 1 else (lambda: sharded.sweep(0))
        edges_per_step_all_ranks = 3 * n_int * world
        config.update({"workload": f"M6-L0-like synthetic level ({lattice}^3 jittered lattice" + (" points, mixed element types" if args.mesh == "mixed" else "") + f", permuted ids): {nel} nodes / "
                                   f"{n_int} internal edges per GPU, flux + update sweep, no MG",
                       "step": "copy, compute_step_factor, 3 x (fluxes, time_step), residual (the reference's per-sweep loops; run as 3 fused launches)",
                       "parallelism": f"{world} mesh copies (the reference's -m {world}), all-reduce(min dt) per sweep" if world > 1 else "1 GPU"})
    elif workload == "partitioned":
        from mgcfd.partition import halo_volume, partition_level, rcb_partition
        mg, levels = build_workload(lattice)
        L = levels[0]
        part = rcb_partition(np.asarray(L["coords"]), world)
        P = partition_level(L, part)[rank]
        solver = make_solver([P.level], mg.mesh_variant, n_owned=[P.n_owned])
        q = perturbed_state(L["nel"], solver.far_field()[:5])
        solver.set(0, "variables", q[P.global_ids])
        nel, n_int = int(L["nel"]), int(L["n_internal"])
        sw = PartitionedSweep(HipSolverAdapter(solver, dev), P, dist,
                              make_buffer=lambda n: torch.empty(max(n, 1), dtype=torch.float64, device=dev), fused=True)
        sw.exchange("variables")
        step = sw.sweep
        exchange = "torch.distributed: a fused launch per stage, pack / batch_isend_irecv / unpack per peer from Python"
        torch_exchange = exchange
        part_candidates = []                                 # ways to run the sweep loop inside the library still to be tried, in order
        part_mode = "torch"
        part_notes = []
        ipc_gave_up = False
        library_passed = [False]                             # the buffered RCCL form reproduced the torch path's sweep at start-up

        def part_reset():
            """every rank back at the perturbed start state, ghosts current (whatever runs the loop)"""
            if world > 1:
                torch.cuda.synchronize(); dist.barrier()      # (nobody is still sweeping, in whatever form, when states are rewritten)
            solver.set(0, "variables", q[P.global_ids])
            if part_mode == "torch":
                sw.exchange("variables")
            else:
                solver.rank_exchange(0)

        def part_try(mode):
            """The sweep loop inside the library (mgcfd_rank_sweeps) — `library`: boundary tiles, one pack, ncclSend/ncclRecv
            grouped on a second stream, the interior tiles under the transfer, one unpack; `ipc`: the messages as direct stores
            into the neighbours' memory (HIP IPC), the time-step all-reduce through the same flags.  Before it is trusted it
            must reproduce, on every rank, the sweep the torch path makes from the same state, bit for bit."""
            nonlocal step, exchange, part_mode, ipc_gave_up
            if mode.startswith("ipc") and ipc_gave_up:
                part_notes.append(f"'{mode}' not tried: a wait for a neighbour's message gave up in another IPC form")
                return False
            try:
                def phase(what, fn):
                    """a step that only THIS rank can see failing: every rank learns of it before the next collective call"""
                    err = None
                    try:
                        out = fn()
                    except Exception as e:
                        out, err = None, e
                    t = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MIN)
                    if float(t.item()) != 1.0:
                        raise RuntimeError(f"{what} failed on some rank" + (f" (here: {err})" if err else ""))
                    return out

                def prepare():
                    solver.rank_ipc_detach(0)               # (whatever was tried before: from closed mappings, the buffered form)
                    solver.set_option("rank_split", {"ipc-unsplit": 0, "ipc-fused": 2}.get(mode, 1))
                    solver.set_option("graph", 1 if (args.rank_graphs and mode == "library") else 0)
                    return solver.rank_ipc_export(0) if mode.startswith("ipc") else None
                blob = phase("preparing the exchange", prepare)
                if mode.startswith("ipc"):
                    blobs = [None] * world
                    dist.all_gather_object(blobs, blob)
                    phase("opening the neighbours' buffers (HIP IPC)", lambda: solver.rank_ipc_attach(0, blobs))   # (every rank's: the time-step all-reduce goes through the flags too)
                part_mode = "torch"; part_reset()
                sw.sweep()
                torch.cuda.synchronize()
                want = solver.get(0, "variables")
                # (the IPC form stores into the NEIGHBOURS' buffers: no rank may start it while another one is still in the torch
                #  path's sweep above — found as an intermittent start-up mismatch in the one-GPU rehearsal; part_reset has the barrier)
                part_mode = mode; part_reset()

                def one_sweep():
                    injected_failure("phase", rank)
                    if injected("hang") and rank == world - 1:         # (rehearsal of the ladder's guard: this rank never comes back)
                        time.sleep(1e6)
                    solver.rank_sweeps(0, 1)
                    torch.cuda.synchronize()
                    # (the status first: while waits that gave up are unacknowledged the library refuses to hand out the state)
                    late_ = solver.rank_ipc_status(0) if mode.startswith("ipc") else 0
                    return solver.get(0, "variables"), late_
                got1, late = phase("the library's sweep", one_sweep)
                same = bool(np.array_equal(got1.view(np.int64), want.view(np.int64))) and late == 0
                if mode.startswith("ipc") and injected_failure("ipc-start"):
                    same = False
                # every rank's verdict, and whether a wait for a neighbour gave up anywhere (about 2 s each: the other IPC forms are then not tried)
                verdict = torch.tensor([1.0 if same else 0.0, -1.0 if late else 0.0], dtype=torch.float64, device=dev)
                dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
                if float(verdict[1].item()) < 0.0:
                    ipc_gave_up = True
                if float(verdict[0].item()) != 1.0:
                    raise RuntimeError("the library's sweep differs from the torch path's on some rank" + (", or a wait for a neighbour's message gave up" if ipc_gave_up else ""))
                step = lambda: solver.rank_sweeps(0, 1)
                if mode == "library":
                    library_passed[0] = True
                info = solver.rank_halo_info(0)
                exchange = (("libmgcfd_hip (mgcfd_rank_sweeps, HIP IPC): one launch per stage stores a rank's nodes into its neighbours' ghost slots and raises "
                             "their flags, the time-step all-reduce through the same flags"
                             + {"ipc": "; boundary tiles, message, interior tiles; ", "ipc-unsplit": "; all tiles in one launch, then the message; ",
                                "ipc-fused": "; ONE launch per stage that sends its own message (boundary tiles first); "}[mode] if mode.startswith("ipc") else
                             "libmgcfd_hip (mgcfd_rank_sweeps): RCCL ncclSend/ncclRecv grouped on a second stream under the interior tiles; ")
                            + f"rank 0: {info['boundary_tiles']} boundary + {info['interior_tiles']} interior tiles; checked against the torch path at start-up"
                            + ("; sweeps replayed from hipGraphs" if (args.rank_graphs and mode == "library") else ""))
                return True
            except Exception as e:
                if rank == 0:
                    print(f"bench.py: exchange '{mode}' not used: {e}", file=sys.stderr)
                part_notes.append(f"'{mode}' not used: {e}")
                if mode.startswith("ipc"):
                    try:
                        solver.rank_ipc_detach(0)
                    except Exception:
                        pass
                part_mode = "torch"
                step, exchange = sw.sweep, torch_exchange
                return False

        def part_next():
            """the next way that passes its start-up check (the torch path when none does)"""
            while part_candidates:
                if part_try(part_candidates.pop(0)):
                    break
            part_reset()

        def part_pick():
            """--exchange auto: every candidate that passes its start-up check — the RCCL form first, then the IPC forms — runs a
            short burst of sweeps; the fastest one is used, the others stay behind it as fall-backs (in the order of their bursts)."""
            timed = []
            for mode in list(part_candidates):
                # (the RCCL form comes FIRST: it is the form whose every call is a documented stream-ordered one; the IPC forms,
                #  whose hand-over between devices has never run across xGMI, are upgrades that must beat it AND pass the end check)
                if part_try(mode):
                    part_reset()
                    for _ in range(3):
                        step()
                    dist.barrier(); torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(12):
                        step()
                    torch.cuda.synchronize()
                    tb = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
                    dist.all_reduce(tb, op=dist.ReduceOp.MAX)
                    timed.append((float(tb.item()) / 12, mode))
                    part_notes.append(f"start-up burst of '{mode}': {timed[-1][0] * 1e6:.0f} us per sweep")
            part_candidates.clear()
            part_candidates.extend(m for _, m in sorted(timed))
            part_next()

        def rank_graphs_leg():
            """Beside the timed form's figure, never as it: the RCCL form's sweeps captured once per buffer rotation and replayed
            from hipGraphs (MGCFD_OPT_GRAPH = 1: one host call per sweep instead of ~25; 10 us of host time against 84 with the one
            rank a one-GPU box offers, tools/hostcost_rccl.py).  ncclSend / ncclRecv inside a capture has never run between two
            devices here, so the form is no candidate of the ladder: it runs after the line's figure is complete, under a guard
            that prints that line, and counts only if (i) the library says sweeps really were replayed and (ii) the state after
            seven sweeps — two passes through the three buffer rotations — equals the torch path's bit for bit on every rank."""
            nonlocal part_mode
            rec = {}
            try:
                def agree(what, fn):
                    err = None
                    try:
                        res = fn()
                    except Exception as e:
                        res, err = None, e
                    t = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MIN)
                    if float(t.item()) != 1.0:
                        raise RuntimeError(f"{what} failed on some rank" + (f" (here: {err})" if err else ""))
                    return res

                def to_library_form():
                    injected_failure("rank-graphs-setup", rank)
                    solver.rank_ipc_detach(0)
                    solver.set_option("rank_split", 1)
                    solver.set_option("graph", 0)
                agree("going back to the buffered RCCL form", to_library_form)
                n_check = 7
                part_mode = "torch"; part_reset()
                for _ in range(n_check):
                    sw.sweep()
                torch.cuda.synchronize()
                want = solver.get(0, "variables")
                part_mode = "library"; part_reset()

                def replayed():
                    injected_failure("rank-graphs-sweeps", rank)
                    solver.set_option("graph", 1)
                    solver.rank_sweeps(0, n_check)
                    torch.cuda.synchronize()
                    return solver.get(0, "variables"), solver.rank_graph_status(0)
                got, status = agree("the captured sweeps", replayed)
                same = bool(np.array_equal(got.view(np.int64), want.view(np.int64)))
                verdict = torch.tensor([1.0 if same else 0.0, 1.0 if (status["sweeps_replayed"] > 0 and not status["capture_refused"]) else 0.0], dtype=torch.float64, device=dev)
                dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
                rec["graph_status_rank0"] = status
                if float(verdict[0].item()) != 1.0:
                    raise RuntimeError(f"the state after {n_check} sweeps differs from the torch path's on some rank")
                if float(verdict[1].item()) != 1.0:
                    raise RuntimeError("a capture was refused on some rank (the sweeps ran call by call, with the right results): no replayed figure")
                for _ in range(args.warmup):
                    solver.rank_sweeps(0, 1)
                torch.cuda.synchronize(); dist.barrier()
                t0g = time.perf_counter()
                for _ in range(args.steps):
                    solver.rank_sweeps(0, 1)
                t_host = time.perf_counter() - t0g
                torch.cuda.synchronize(); dist.barrier()
                tg = torch.tensor([time.perf_counter() - t0g, t_host], dtype=torch.float64, device=dev)
                dist.all_reduce(tg, op=dist.ReduceOp.MAX)
                e = float(tg[0].item())
                rec.update({"us_per_sweep": round(e / max(args.steps, 1) * 1e6, 3), "host_us_per_sweep": round(float(tg[1].item()) / max(args.steps, 1) * 1e6, 3),
                            "medges_per_s": round(3 * int(L["n_internal"]) * args.steps / e / 1e6, 3), "steps": args.steps, "warmup": args.warmup,
                            "graph_status_rank0": solver.rank_graph_status(0),
                            "what": "mgcfd_rank_sweeps with MGCFD_OPT_GRAPH = 1: every sweep of an RCCL rank (boundary tiles, pack, grouped ncclSend / ncclRecv on a "
                                    "second stream, interior tiles, unpack, the all-reduce of the time step) replayed from a hipGraph captured once per buffer "
                                    f"rotation; the state after {n_check} sweeps equals the torch path's bit for bit on every rank; max over ranks, barriers on both sides"})
            except Exception as e:
                rec["error"] = f"{type(e).__name__}: {e}"
            finally:
                try:
                    solver.set_option("graph", 0)
                except Exception:
                    pass
            return rec

        ladder_guard = None
        if world == 1:
            step, exchange = (lambda: solver.smooth(0, 1)), None
        elif args.exchange != "torch":
            # Before anything of the library's own loops runs between ranks (none of it has run on several GPUs yet): the torch
            # path's figure over the same W + K sweeps, kept as the line to print if a later rung never comes back — a rank stuck
            # inside a collective cannot be interrupted, so a guard prints that line and every rank leaves (leg_guard).
            part_reset()
            for _ in range(args.warmup):
                sw.sweep()
            torch.cuda.synchronize(); dist.barrier()
            t0f = time.perf_counter()
            for _ in range(args.steps):
                sw.sweep()
            torch.cuda.synchronize()
            tf = torch.tensor([time.perf_counter() - t0f], dtype=torch.float64, device=dev)
            dist.all_reduce(tf, op=dist.ReduceOp.MAX)
            fallback_elapsed = float(tf.item())
            part_notes.append(f"torch path over the same {args.warmup} + {args.steps} sweeps, measured first: {fallback_elapsed / max(args.steps, 1) * 1e6:.0f} us per sweep")
            part_reset()

            def fallback_line():
                return {"metric": "Medges/s (compute_flux_edge)", "value": round(3 * int(L["n_internal"]) * args.steps / fallback_elapsed / 1e6, 3), "unit": "Medges/s",
                        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(fallback_elapsed / max(args.steps, 1) * 1e3, 6),
                        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                        "config": {"workload": f"M6-L0-like level tiled 8x ({lattice}^3 jittered lattice): {int(L['nel'])} nodes / {int(L['n_internal'])} internal edges, recursive coordinate bisection into {world} parts",
                                   "workload_kind": "partitioned", "collectives": backend, "numerics": config.get("numerics"),
                                   "exchange": torch_exchange + " (the figure of the torch path measured first: a form of the library's own loop did not come back; " + "; ".join(part_notes) + ")"}}
            ladder_guard = leg_guard(float(os.environ.get("MGCFD_BENCH_LADDER_S", "420")), rank, fallback_line, "library_forms", "the library's rank loop (start-up checks or timed sweeps)")
            ladder_guard.__enter__()
            setup_error = None
            uid = [None]
            if not rehearsal:
                try:
                    uid = [mgcfd.rccl_unique_id() if rank == 0 else None]
                except Exception as e:                       # (rank 0 only can fail here; it still takes part in the broadcast)
                    setup_error = e
                dist.broadcast_object_list(uid, src=0)
            try:
                if setup_error:
                    raise setup_error
                injected_failure("attach", rank)
                if rehearsal:
                    # (every rank on device 0: RCCL refuses that; the IPC form needs no collective library at all)
                    solver.rank_attach_plain(rank, world)
                    part_candidates = ["ipc-fused", "ipc", "ipc-unsplit"] if args.exchange == "auto" else (["ipc"] if args.exchange == "ipc" else [])
                else:
                    if uid[0] is None:
                        raise RuntimeError("no RCCL unique id from rank 0")
                    solver.rank_attach_rccl(rank, world, uid[0])
                    part_candidates = ["library", "ipc-fused", "ipc", "ipc-unsplit"] if args.exchange == "auto" else [args.exchange]
                solver.rank_set_halo(0, P)
            except Exception as e:
                setup_error = e
            # (a failure only one rank sees must not leave the others waiting in the next collective call)
            t_ok = torch.tensor([0.0 if setup_error else 1.0], dtype=torch.float64, device=dev)
            dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
            if float(t_ok.item()) != 1.0:
                if rank == 0:
                    print(f"bench.py: the library's rank loop not used: {setup_error or 'it failed on another rank'}", file=sys.stderr)
                part_notes.append(f"the library's rank loop not used: {setup_error or 'it failed on another rank'}")
                part_candidates = []
            if args.exchange == "auto" and len(part_candidates) > 1:
                part_pick()
            else:
                part_next()
        scaling = "strong"
        edges_per_step_all_ranks = 3 * n_int
        halo_nodes = int(halo_volume(L, part)) if world > 1 else 0
        try:
            extra["collectives_as_the_library_sees_them"] = solver.rank_info()    # (ranks, transport, ncclCommCount: not what the launcher said)
        except Exception as e:
            extra["collectives_as_the_library_sees_them"] = {"error": str(e)}
        config.update({"workload": f"M6-L0-like level tiled 8x ({lattice}^3 jittered lattice, permuted ids, connected): {nel} nodes / {n_int} "
                                   f"internal edges in total, recursive coordinate bisection into {world} part(s)",
                       "step": "compute_step_factor + all-reduce(min dt), 3 x (fused fluxes + time_step launch, halo message to every neighbouring rank), residual",
                       "parallelism": f"{world} ranks, owner computes, ghosts read-only", "halo_nodes_total": halo_nodes,
                       "halo_bytes_per_stage_all_ranks": halo_nodes * 40, "halo_bytes_per_stage_rank0_sent": int(sum(len(v) for v in P.send.values())) * 40,
                       "library_ranks": extra.get("collectives_as_the_library_sees_them"), "peers_of_rank0": len(sw.peers),
                       "owned_nodes_rank0": int(P.n_owned), "local_internal_edges_rank0": int(P.level["n_internal"]), "exchange": exchange})
    else:                                                    # level-per-gpu
        mg, levels = build_hierarchy()
        solver = make_solver(levels, mg.mesh_variant)
        nlev = solver.num_levels
        cyc = LevelPerRankCycle(HipSolverAdapter(solver, dev), nlev, rank, world, dist=dist)
        step = cyc.cycle if world > 1 else (lambda: solver.run_cycles(1))
        n_ints = [solver.num_internal_edges(l) for l in range(nlev)]
        # a V-cycle sweeps levels 0..n-1 then n-2..1: every level but the first and the last twice
        edges_per_step_all_ranks = 3 * (n_ints[0] + n_ints[-1] + 2 * sum(n_ints[1:-1]))
        scaling = "strong"
        nel, n_int = solver.nel(0), n_ints[0]
        config.update({"workload": f"4-level M6-like synthetic hierarchy {[solver.nel(l) for l in range(nlev)]} nodes, level l on rank l % {world}",
                       "step": "one multigrid V-cycle (sweeps on levels 0,1,2,3,2,1; 3 restrictions, 3 prolongations); restricted variables and coarse "
                               "residuals cross ranks as whole-array point-to-point messages",
                       "parallelism": f"{world} ranks, one level per rank (placement, not concurrency: the V-cycle is sequential in levels)"})

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    flux_only = probe = stream_ceiling = two_in_flight = None
    preheat_sweeps = 0
    if workload in ("level0", "copies"):
        # the standalone compute_flux_edge kernel (writes fluxes[]), 50 back-to-back launches between one hipEvent pair
        # on the same stream, and the data-movement probe on the same tiles (indirect_rw: same loads and stores, the
        # reference's trivial arithmetic) = the empirical ceiling of this layout (src/Kernels/indirect_rw_loop.cpp:8-10).
        # Measured BEFORE the timed region, 1000 launches each (~35 ms of GPU work together): they leave the state alone, and an
        # MI355X that has been idle takes tens of milliseconds of load before its clocks are up — the first 20 sweeps after an
        # idle period run at 66-72 us each, the same 20 sweeps after 30 ms of load at 61.5 (tools/exp/first_steps.py); the
        # driver's default of 5 warm-up steps is 0.3 ms.  The line says so in "preheat".
        # (first of all — it builds and frees a second solver, half a second of idle device — what the kernel does with two
        #  independent launches in flight: roofline.two_launches_in_flight)
        if workload == "level0" and world == 1 and args.variant == -1 and not args.no_two_in_flight and os.environ.get("MGCFD_BENCH_NO_TWO_IN_FLIGHT") != "1":
            two_in_flight = two_launches_in_flight(solver, levels, mg.mesh_variant, local_rank, args.fast)
        solver.bench_flux(0, 2 * ROOFLINE_LAUNCHES)             # (untimed: the ramp itself)
        flux_only = solver.bench_flux(0, ROOFLINE_LAUNCHES)
        flux_contracted = flux_free = None
        if not args.fast:
            # the same launch with FMA contraction allowed (MGCFD_OPT_EXACT = 0: results within 1e-12 relative of the reference's,
            # tests/test_gpu_parity.py REL_FAST; north_star's bound is 1e-10) — reported beside the bit-identical figure, never as it
            solver.set_option("exact", 0)
            if args.variant == -1:
                solver.set_option("flux_variant", 1)                # (the node gather: auto would take the order-free kernel in this mode)
            solver.bench_flux(0, ROOFLINE_LAUNCHES)                 # (untimed: the first batch after the switch reads 1 % slower)
            flux_contracted = solver.bench_flux(0, ROOFLINE_LAUNCHES)
            # ... and with ORDER-FREE accumulation on top (k_flux_free, variant bit 6, the contracted namespace only: every edge of a
            # tile evaluated once, the other end's share added to its LDS sum with ds_add_f64; tests/test_gpu_order_free.py)
            if args.variant == -1 and solver.has_order_free(0):
                solver.set_option("flux_variant", 65)
                solver.bench_flux(0, ROOFLINE_LAUNCHES)
                flux_free = solver.bench_flux(0, ROOFLINE_LAUNCHES)
                solver.set_option("flux_variant", args.variant)
            solver.set_option("exact", 1)
        probe = solver.bench_indirect_rw(0, ROOFLINE_LAUNCHES) if hasattr(solver, "bench_indirect_rw") else None
        # ... and a tile-shaped STREAM of exactly the algorithmic bytes (one workgroup per tile, nothing dependent, nothing computed)
        stream_ceiling = solver.bench_stream_ceiling(0, ROOFLINE_LAUNCHES)
        solver.zero_fluxes(0)                                   # (the sweeps start from zero fluxes, as after any time_step)
        # A short timed region (the driver's 20 steps are 1.1 ms) would otherwise run before the device has settled under THIS
        # kernel: after the flux launches above the first few hundred sweeps still take 57.6-60 us against 55.2 in the steady state
        # (same box, --steps 20 ... 400 against 2,000).  So, when fewer than 1,000 steps are timed, 1,000 sweeps of the workload itself
        # (55 ms) run untimed in front of the W warm-up steps; the timed steps are the same operations on a later state.
        if workload == "level0" and world == 1 and args.steps < 1000 and os.environ.get("MGCFD_BENCH_NO_PREHEAT_SWEEPS") != "1":
            solver.smooth(0, PREHEAT_SWEEPS)
            preheat_sweeps = PREHEAT_SWEEPS
    while True:
        for _ in range(args.warmup):
            step()
        barrier()
        solver.reset_monitoring()
        # ONE hipEvent pair around the whole timed region, on the stream the launches go to (no event inside the region:
        # on ROCm an event record is a barrier packet between two launches).  A level0 step is exactly three fused-stage
        # launches back to back, so (event time) / 3K is that kernel's launch-to-launch duration over the timed region.
        live_timing = os.environ.get("MGCFD_BENCH_NO_TIMING") != "1" and workload in ("level0", "copies")
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        import gc
        gc_was_on = gc.isenabled()
        gc.disable()                                            # (a collection inside a 1.1 ms region would be most of it)
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(args.steps):
            step()
        ev1.record(stream)
        barrier()
        elapsed = time.perf_counter() - t0
        if gc_was_on:
            gc.enable()
        gpu_seconds = ev0.elapsed_time(ev1) * 1e-3
        if not (workload == "partitioned" and world > 1 and part_mode.startswith("ipc")):
            break
        # The IPC exchange has only ever been rehearsed on ONE GPU, so its figure counts only if the state it leaves is the
        # state the torch path leaves after the same W + K sweeps from the same start, bit for bit on every rank, and no wait
        # for a neighbour gave up; otherwise the next way runs and is timed instead.
        late = solver.rank_ipc_status(0)                    # (first: it acknowledges; the library refuses the state while it is unacknowledged)
        got = solver.get(0, "variables")
        part_mode_run = part_mode
        part_mode = "torch"; part_reset()
        for _ in range(args.warmup + args.steps):
            sw.sweep()
        torch.cuda.synchronize()
        ref_final = solver.get(0, "variables")
        n_diff = int(np.count_nonzero(np.any(got.view(np.int64) != ref_final.view(np.int64), axis=1)))
        same = late == 0 and n_diff == 0 and not injected_failure("ipc-end")
        if not same:
            print(f"bench.py: rank {rank}: IPC run: {late} wait(s) gave up, {n_diff} node(s) differ from the torch path's final state "
                  f"({int(np.count_nonzero(np.any(got[:P.n_owned].view(np.int64) != ref_final[:P.n_owned].view(np.int64), axis=1)))} owned)", file=sys.stderr)
        ok = torch.tensor([1.0 if same else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) == 1.0:
            exchange += f"; the state after the {args.warmup} + {args.steps} sweeps equals the torch path's bit for bit on every rank"
            break
        if rank == 0:
            print("bench.py: the IPC exchange left a different state than the torch path (or a wait gave up): its figure is discarded", file=sys.stderr)
        part_notes.append(f"'{part_mode_run}' discarded after the run: its final state differed from the torch path's, or a wait for a neighbour gave up")
        try:
            solver.rank_ipc_detach(0)
        except Exception:
            pass
        step, exchange = sw.sweep, torch_exchange
        part_next()
    if workload == "partitioned":
        if ladder_guard is not None:
            ladder_guard.cancel()                                # (the sweeps were measured: the legs that follow have guards of their own)
        config["exchange"] = exchange + ("".join(f" ({n})" for n in part_notes) if world > 1 else "") if exchange else exchange
    flux_launches = 3 * args.steps if live_timing and workload == "level0" else 0
    flux_avg = gpu_seconds / flux_launches if flux_launches else 0.0
    rc, bad = solver.check_for_invalid_variables(0)
    if rc != 0:
        raise SystemExit(f"state became invalid during the bench (code {rc}, cell {bad})")

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    out = None
    if rank == 0:
        out = {
            "metric": "Medges/s (compute_flux_edge)" + (" + MG V-cycle wall-s" if workload == "level0" and not args.no_vcycle else ""),
            "value": round(edges_per_step_all_ranks * args.steps / elapsed / 1e6, 3),
            "unit": "Medges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 6),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": dict(config, workload_kind=workload, collectives=backend),
        }
        if workload == "level-per-gpu":
            out["vcycle"] = {"wall_s_per_cycle": round(elapsed / args.steps, 9), "flux_edge_iterations_per_cycle": edges_per_step_all_ranks}
    if workload in ("level0", "copies"):
        if rank == 0:
            bytes_flux = 40 * n_int + 80 * nel                  # compute_flux_edge
            bytes_ts = 168 * nel                                # time_step
            traffic = {}
            try:
                if lattice == LATTICE and args.variant == -1 and args.mesh == "lattice":
                    traffic = json.load(open(os.path.join(ROOT, TRAFFIC_PROFILE)))
            except (OSError, ValueError):
                traffic = {}
            order_free_here = bool(args.fast and args.variant == -1 and hasattr(solver, "has_order_free") and solver.has_order_free(0))
            ach = bytes_flux / flux_only / 1e9 if flux_only > 0 else 0.0
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    # (--fast: MGCFD_OPT_EXACT = 0 launches the order-free kernel for this loop where the level has its plan)
                    "traffic": (traffic.get("flux_order_free" if (args.fast and order_free_here) else "flux_only") or {}).get("bytes") if not (args.fast and not order_free_here) else None,
                    "traffic_source": (f"{TRAFFIC_PROFILE} (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes; build {traffic.get('build')})"
                                       if traffic else None),
                    "frac_of_measured_copy_rate": round(ach / HBM_COPY_GBS, 4), "measured_copy_rate": HBM_COPY_GBS,
                    "kernel": "compute_flux_edge + boundary + far-field faces in one launch (writes fluxes[], no time_step): the kernel BASELINE's 60 % target names"
                              + ("; --fast: the order-free kernel k_flux_free (<= 1e-12 per launch, tests/test_gpu_order_free.py), not the bit-identical one" if (args.fast and order_free_here) else ""),
                    "launches": ROOFLINE_LAUNCHES, "avg_kernel_us": round(flux_only * 1e6, 3), "algorithmic_bytes_per_launch": bytes_flux,
                    "algorithmic_bytes": "40 B per internal edge + 80 B per node (SURVEY.md §8d)",
                    "medges_per_s": round(n_int / flux_only / 1e6, 1) if flux_only > 0 else None}
            roof["preheat"] = (f"the {(3 if args.fast else (7 if flux_free else 5)) * ROOFLINE_LAUNCHES} flux + {ROOFLINE_LAUNCHES} probe launches above ({2 if args.fast else (4 if flux_free else 3)} x {ROOFLINE_LAUNCHES} of them untimed) ran BEFORE the warm-up and timed steps "
                               "(an idle MI355X needs ~30 ms of load before its clocks are up: tools/exp/first_steps.py)"
                               + (f"; then {preheat_sweeps} untimed sweeps of the workload itself in front of the {args.warmup} warm-up steps "
                                  "(fewer than 1,000 timed steps: the device settles under the stage kernel over tens of milliseconds)" if preheat_sweeps else ""))
            if roof["traffic"]:
                roof["traffic_over_algorithmic"] = round(roof["traffic"] / bytes_flux, 3)
            if stream_ceiling:
                roof["practical_ceiling_us"] = round(stream_ceiling * 1e6, 3)
                roof["practical_ceiling"] = (f"a tile-shaped stream of exactly the {bytes_flux} algorithmic bytes ({n_int} x 40 + {nel} x 40 read, {nel} x 40 written; one workgroup per tile, "
                                             "16-byte loads, eight in flight, write-through stores, the order-free kernel's LDS footprint; nothing dependent, nothing computed), "
                                             f"{ROOFLINE_LAUNCHES} back-to-back launches between one hipEvent pair, launch boundary included (mgcfd_bench_stream_ceiling)")
                roof["practical_ceiling_frac"] = round(bytes_flux / stream_ceiling / 1e9 / HBM_PEAK_GBS, 4)
                roof["flux_over_practical_ceiling"] = round(flux_only / stream_ceiling, 3)
            if two_in_flight:
                for rec in two_in_flight.values():
                    if isinstance(rec, dict) and rec.get("us_per_launch"):
                        rec["frac"] = round(bytes_flux / (rec["us_per_launch"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                roof["two_launches_in_flight"] = two_in_flight
            if probe:
                roof["empirical_ceiling_us"] = round(probe * 1e6, 3)
                roof["empirical_ceiling"] = "indirect_rw through the same tiles (same loads and stores, the reference's trivial arithmetic; src/Kernels/indirect_rw_loop.cpp:8-10)"
                roof["flux_over_indirect_rw"] = round(flux_only / probe, 3)
            if flux_contracted:
                roof["fma_contracted"] = {"avg_kernel_us": round(flux_contracted * 1e6, 3),
                                          "frac": round(bytes_flux / flux_contracted / 1e9 / HBM_PEAK_GBS, 4),
                                          "numerics": "the same launch with FMA contraction allowed (MGCFD_OPT_EXACT = 0): within 1e-12 relative of the reference "
                                                      "(north_star allows 1e-10); the line's other figures are the bit-identical mode"}
            if flux_free:
                roof["order_free"] = {"avg_kernel_us": round(flux_free * 1e6, 3), "launches": ROOFLINE_LAUNCHES,
                                      "traffic": traffic.get("flux_order_free", {}).get("bytes"),
                                      "traffic_over_algorithmic": round(traffic["flux_order_free"]["bytes"] / bytes_flux, 3) if traffic.get("flux_order_free", {}).get("bytes") else None,
                                      "frac": round(bytes_flux / flux_free / 1e9 / HBM_PEAK_GBS, 4),
                                      "medges_per_s": round(n_int / flux_free / 1e6, 1),
                                      "kernel": "k_flux_free (MGCFD_OPT_EXACT = 0, MGCFD_OPT_FLUX_VARIANT = 65): one 28-byte entry per edge and tile, -F added to the other "
                                                "end's LDS sum with ds_add_f64, one barrier, 56-byte records, four workgroups per CU",
                                      "numerics": "sums associated differently from the reference's and not reproducible bit for bit: <= 1e-12 relative per launch and "
                                                  "sweep, <= 1e-10 on level-0 variables after 25 full-size V-cycles, the reference's -v rule passes "
                                                  "(tests/test_gpu_order_free.py); what MGCFD_OPT_EXACT = 0 launches for this loop — the line's other figures are the bit-identical mode"}
            if flux_avg > 0:
                a2 = (bytes_flux + bytes_ts) / flux_avg / 1e9
                # (--fast: the sweeps run the role-specialised order-free stages where the level has the plan)
                stage_traffic = ((traffic.get("fused_stage_order_free") if (args.fast and order_free_here) else traffic.get("fused_stage")) or {}).get("bytes")
                roof["fused_stage"] = {"kernel": "one whole Runge-Kutta stage per launch = compute_flux_edge + boundary + far-field + time_step: what the timed sweeps run",
                                       "launches": flux_launches, "avg_kernel_us": round(flux_avg * 1e6, 3),
                                       "timed_by": "one hipEvent pair around the K timed steps (3 launches each, back to back) on the launch stream",
                                       "algorithmic_bytes_per_launch": bytes_flux + bytes_ts,
                                       "algorithmic_bytes": {"compute_flux_edge (40E+80N)": bytes_flux, "time_step (168N)": bytes_ts},
                                       "achieved": round(a2, 1), "frac": round(a2 / HBM_PEAK_GBS, 4),
                                       "frac_note": "priced as the TWO loops it replaces (the flux written and read again: 40 N more than a fused launch moves); "
                                                    "bytes_a_fused_stage_must_move prices what fusion leaves",
                                       "bytes_a_fused_stage_must_move": 40 * n_int + 128 * nel,
                                       "bytes_a_fused_stage_must_move_terms": "40 B per internal edge + 128 B per node: the stage's input state 40, the sweep's start state 40, "
                                                                              "the new state 40, step factor or volume 8 (the flux never leaves registers)",
                                       "frac_of_bytes_a_fused_stage_must_move": round((40 * n_int + 128 * nel) / flux_avg / 1e9 / HBM_PEAK_GBS, 4),
                                       "frac_if_priced_as_flux_only": round(bytes_flux / flux_avg / 1e9 / HBM_PEAK_GBS, 4),
                                       "traffic": stage_traffic,
                                       "traffic_over_bytes_a_fused_stage_must_move": round(stage_traffic / (40 * n_int + 128 * nel), 3) if stage_traffic else None}
            out["roofline"] = roof
    if workload == "partitioned" and world > 1 and not args.no_rank_graphs and not args.rank_graphs and (library_passed[0] or (rehearsal and injected("rank-graphs-rehearsal"))):
        with leg_guard(float(os.environ.get("MGCFD_BENCH_LEG_S", "420")), rank, out, "rank_graphs", "the RCCL ranks' sweeps replayed from hipGraphs"):
            graphs_rec = rank_graphs_leg()
        if rank == 0:
            out["rank_graphs"] = graphs_rec
    solver.close()
    if rank == 0 and world == 1 and workload == "level0" and args.mesh == "lattice" and lattice == LATTICE and not args.no_vcycle and "roofline" in out:
        out["roofline"]["mixed_mesh"] = mixed_mesh_roofline(make_solver, args.fast)
    if world > 1:
        dist.barrier()
    vc_sizes = {"8x": HIERARCHY_8X, "base": HIERARCHY, "tiny": (20, 12, 8, 6)}[args.vcycle_hierarchy]
    kept = [] if (rank == 0 and not args.no_group) else None
    if world > 1 and workload in ("partitioned", "partitioned-mg") and not args.no_vcycle:
        # the second half of BASELINE's metric on N GPUs: every level of the hierarchy partitioned, the cycle inside the library.
        # (guarded: nothing of it has run on several GPUs yet, and the sweeps' figure must survive whatever it does)
        with leg_guard(float(os.environ.get("MGCFD_BENCH_LEG_S", "420")), rank, out, "vcycle", "the V-cycle leg on the partitioned hierarchy"):
            vc = vcycle_partitioned(args, dist, world, rank, dev, stream, rehearsal, vc_sizes, keep=kept)
        if rank == 0:
            out["vcycle"] = vc
            out["metric"] = "Medges/s (compute_flux_edge) + MG V-cycle wall-s"
    if world > 1 and workload in ("partitioned", "partitioned-mg") and not args.no_group:
        # beside the RCCL ranks' figure, never as it: the SAME level and hierarchy swept by ONE process over all N devices (the
        # in-process group the drop-in binary runs).  Rank 0 runs it while the other ranks — their solvers closed — wait on the
        # rendezvous store, not in a collective (a barrier kernel would spin on the devices being measured).
        from torch.distributed.distributed_c10d import _get_default_store
        store = _get_default_store()
        with leg_guard(float(os.environ.get("MGCFD_BENCH_LEG_S", "420")), rank, out, "in_process_group", "the in-process group leg"):
            torch.cuda.synchronize()
            dist.barrier()
            if rank == 0:
                try:
                    torch.cuda.empty_cache()
                    out["in_process_group"] = in_process_group_leg(args, world, lattice, (mg, levels) if levels is not None else None, None if args.no_vcycle else vc_sizes,
                                                                   kept[0] if kept else None, rehearsal, args.steps, args.warmup)
                except Exception as e:
                    out["in_process_group"] = {"error": f"{type(e).__name__}: {e}"}
                store.set("mgcfd_group_leg_done", "1")
            else:
                import datetime
                store.wait(["mgcfd_group_leg_done"], datetime.timedelta(seconds=float(os.environ.get("MGCFD_BENCH_LEG_S", "420")) + 120.0))
    if rank == 0:
        if world == 1 and workload == "level0" and not args.no_vcycle:
            out["vcycle"] = vcycle_wall(args.fast, device=local_rank)
        if args.cpu_seconds > 0 and world == 1 and workload == "level0":
            out["cpu_baseline"] = cpu_baseline(levels, args.cpu_seconds)
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()
        faulthandler.cancel_dump_traceback_later()


def plumbing_only(args, dist, world, rank, workload, lattice):
    """No GPU: rendezvous over gloo, build this rank's share of the host-side partition (small lattice), agree on the
    maximum of a timed no-op loop exactly as the real run does, print the line with value null."""
    import torch
    if world > 1:
        with quiet_stdout():
            dist.init_process_group("gloo", rank=rank, world_size=world)
    detail = {}
    if workload == "partitioned":
        from mgcfd.partition import halo_volume, partition_level, rcb_partition
        mg, levels = build_workload(args.lattice or 12)
        part = rcb_partition(np.asarray(levels[0]["coords"]), world)
        P = partition_level(levels[0], part)[rank]
        owned = torch.tensor([P.n_owned], dtype=torch.int64)
        if world > 1:
            dist.all_reduce(owned)
        assert int(owned.item()) == int(levels[0]["nel"]), "the parts must cover the level exactly once"
        detail = {"nodes": int(levels[0]["nel"]), "halo_nodes_total": int(halo_volume(levels[0], part)) if world > 1 else 0}
    t = torch.tensor([0.0], dtype=torch.float64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n = dist.get_world_size()
        dist.destroy_process_group()
    else:
        n = 1
    if rank == 0:
        print(json.dumps({"metric": "Medges/s (compute_flux_edge)", "value": None, "unit": "Medges/s", "n_gpus": n, "steps": args.steps,
                          "warmup": args.warmup, "plumbing_only": True, "config": dict(detail, workload_kind=workload)}))


if __name__ == "__main__":
    main()
