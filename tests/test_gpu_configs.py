"""BASELINE.json's configurations at their FULL sizes on the GPU, against the oracle, bit for bit.

configs[0] fvcorr-like 97,335 nodes, 1,000 iterations          test_cfg1_*
configs[1] M6-L0-like 300,763 nodes: the sweeps the bench times  test_cfg2_*
configs[2] 4-level hierarchy 300,763/166,375/110,592/79,507     test_cfg3_*
configs[4] the level tiled 8x (2.4 M nodes) split 8 ways         test_cfg5_*  (8 solvers on this one GPU, threads for ranks)
configs[3] (one level per GPU) at full size is the same arithmetic as configs[2]; its in-process analogue at full size:
                                                                 test_cfg4_*
"""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits_equal(got, want, what):
    got, want = np.ascontiguousarray(got), np.ascontiguousarray(want)
    bad = np.flatnonzero(got.ravel().view(np.int64) != want.ravel().view(np.int64))
    assert bad.size == 0, f"{what}: {bad.size} values differ bitwise; first {bad[:5]}, max abs diff {np.abs(got - want).max():.3e}"


def _oracle_levels(oracle, levels):
    """OraLevel array over in-memory level dicts + the arrays that must stay alive."""
    n = len(levels)
    lv = (oracle.OraLevel * n)()
    keep = []
    for l, L in enumerate(levels):
        vol = np.ascontiguousarray(L["volumes"], dtype=np.float64)
        coords = np.ascontiguousarray(L["coords"], dtype=np.float64)
        edges = np.ascontiguousarray(L["edges"]).copy()
        state = [np.zeros((L["nel"], 5)) for _ in range(4)] + [np.zeros(L["nel"])]
        m = None
        lv[l].nel, lv[l].n_edges = L["nel"], len(edges)
        lv[l].n_internal, lv[l].n_boundary, lv[l].n_wall = L["n_internal"], L["n_boundary"], L["n_wall"]
        lv[l].internal_start, lv[l].boundary_start, lv[l].wall_start = 0, L["n_internal"], L["n_internal"] + L["n_boundary"]
        lv[l].volumes, lv[l].coords, lv[l].edges = oracle.ptr(vol), oracle.ptr(coords), oracle.ptr(edges)
        lv[l].variables, lv[l].old_variables, lv[l].residuals, lv[l].fluxes = (oracle.ptr(a) for a in state[:4])
        lv[l].step_factors = oracle.ptr(state[4])
        if L.get("mg_map") is not None and l + 1 < n:
            m = np.ascontiguousarray(L["mg_map"], dtype=np.int64)
            lv[l].mg_map, lv[l].mgc = oracle.ptr(m), len(m)
        keep.append(dict(vol=vol, coords=coords, edges=edges, variables=state[0], old=state[1], residuals=state[2],
                         fluxes=state[3], sf=state[4], map=m))
    return lv, keep


def test_cfg3_full_size_four_level_vcycles(oracle):
    """BASELINE configs[2] at size: the (67, 55, 48, 43)^3 hierarchy bench.py's V-cycle leg times, 2 V-cycles from the
    far field: every level's variables and residuals and the loop counters against ora_solve."""
    import bench
    import mgcfd
    mg, levels = bench.build_hierarchy()
    assert [L["nel"] for L in levels] == [300763, 166375, 110592, 79507]
    cycles = 2
    lv, keep = _oracle_levels(oracle, levels)
    lib = oracle.load()
    want_rms = np.zeros(cycles)
    iters = (oracle.OraIters * len(levels))()
    assert lib.ora_solve(lv, len(levels), mg.mesh_variant, cycles, 0, oracle.ptr(want_rms), iters) == 0
    s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    rms = s.run_cycles(cycles)
    for l in range(len(levels)):
        _bits_equal(s.get(l, "variables"), keep[l]["variables"], f"level {l} variables after {cycles} V-cycles")
        _bits_equal(s.get(l, "residuals"), keep[l]["residuals"], f"level {l} residuals")
        got = s.loop_iters(l)
        want = {"flux": iters[l].flux, "update": 0, "compute_step": iters[l].compute_step, "time_step": iters[l].time_step,
                "restrict": iters[l].restrict_, "prolong": iters[l].prolong, "indirect_rw": 0}
        assert got == want, f"level {l} LoopNumIters: {got} vs {want}"
    assert np.allclose(rms, want_rms, rtol=1e-12, atol=0)
    # the same cycles with one launch per loop (what the driver's per-loop timers run) and replayed from hipGraphs
    for opts in (dict(fuse_update=0, graph=0), dict(fuse_update=1, graph=1)):
        s2 = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
        for k, v in opts.items():
            s2.set_option(k, v)
        s2.run_cycles(cycles)
        for l in range(len(levels)):
            _bits_equal(s2.get(l, "variables"), keep[l]["variables"], f"{opts}: level {l}")
        s2.close()
    s.close()


def test_cfg2_full_size_sweeps_the_bench_times(oracle):
    """BASELINE configs[1] at size, the path bench.py times: 4 consecutive fused sweeps on the 300,763-node level from
    the bench's perturbed state — the first takes the k_step_factor_local branch, the next three the look-ahead one
    (the last stage of sweep n leaves sweep n+1's step-factor minima) — against the oracle's loops, sweep by sweep."""
    import bench
    import mgcfd
    mg, levels = bench.build_workload(67)
    L = levels[0]
    lib = oracle.load()
    edges = np.ascontiguousarray(L["edges"]).copy()
    coords = np.ascontiguousarray(L["coords"], dtype=np.float64)
    lib.ora_adjust_ewt(oracle.ptr(coords), len(edges), oracle.ptr(edges))
    lib.ora_dampen_ewt(len(edges), oracle.ptr(edges), 5e-8)
    ff = oracle.farfield()
    nel, ni, nb, nw = L["nel"], L["n_internal"], L["n_boundary"], L["n_wall"]
    vol = np.ascontiguousarray(L["volumes"], dtype=np.float64)
    s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    q = bench.perturbed_state(nel, s.far_field()[:5])
    s.set(0, "variables", q)
    v, f, sf = q.copy(), np.zeros_like(q), np.zeros(nel)
    for sweep in range(4):
        old = v.copy()
        lib.ora_compute_step_factor(nel, oracle.ptr(v), oracle.ptr(vol), oracle.ptr(sf))
        for j in range(3):
            lib.ora_compute_flux_edge(0, ni, oracle.ptr(edges), oracle.ptr(v), oracle.ptr(f))
            lib.ora_compute_boundary_flux_edge(ni, nb, oracle.ptr(edges), oracle.ptr(v), oracle.ptr(f))
            lib.ora_compute_wall_flux_edge(ni + nb, nw, oracle.ptr(edges), oracle.ptr(v), oracle.ptr(f), C.byref(ff))
            lib.ora_time_step(j, nel, oracle.ptr(sf), oracle.ptr(f), oracle.ptr(old), oracle.ptr(v))
        s.smooth(0, 1)
        _bits_equal(s.get(0, "variables"), v, f"sweep {sweep}: variables")
        _bits_equal(s.get(0, "step_factors"), sf, f"sweep {sweep}: step factors")
        _bits_equal(s.get(0, "old_variables"), old, f"sweep {sweep}: old_variables")
        _bits_equal(s.get(0, "residuals"), v - old, f"sweep {sweep}: residuals")
    # ... and the standalone compute_flux_edge kernel the roofline figure is quoted on, plus the indirect_rw probe
    s.zero_fluxes(0)
    s.compute_flux_edge(0)
    f[:] = 0.0
    lib.ora_compute_flux_edge(0, ni, oracle.ptr(edges), oracle.ptr(v), oracle.ptr(f))
    _bits_equal(s.get(0, "fluxes"), f, "compute_flux_edge at size")
    s.indirect_rw(0)
    lib.ora_indirect_rw(0, ni, oracle.ptr(edges), oracle.ptr(v), oracle.ptr(f))
    _bits_equal(s.get(0, "fluxes"), f, "indirect_rw at size (through the LDS tiles)")
    s.close()


def test_mixed_level_full_size_sweep(oracle):
    """The non-uniform-degree headline level (bench.py --mesh mixed: hexahedral core, prism layers on a wall, tetrahedral far
    field on the 67^3 points: 300,763 nodes / 1,004,901 internal edges, internal degrees 3 ... 14) at size: the standalone
    flux launch and two fused sweeps against the oracle's loops, bit for bit."""
    import bench
    import mgcfd
    mg, levels = bench.build_workload(67, mesh="mixed")
    L = levels[0]
    assert L["nel"] == 300763 and L["n_internal"] == 1004901
    lib = oracle.load()
    edges = np.ascontiguousarray(L["edges"]).copy()
    coords = np.ascontiguousarray(L["coords"], dtype=np.float64)
    lib.ora_adjust_ewt(oracle.ptr(coords), len(edges), oracle.ptr(edges))
    lib.ora_dampen_ewt(len(edges), oracle.ptr(edges), 5e-8)
    ff = oracle.farfield()
    nel, ni, nb, nw = L["nel"], L["n_internal"], L["n_boundary"], L["n_wall"]
    vol = np.ascontiguousarray(L["volumes"], dtype=np.float64)
    s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    q = bench.perturbed_state(nel, s.far_field()[:5])
    s.set(0, "variables", q)
    s.zero_fluxes(0)
    s.compute_fluxes(0)
    f = np.zeros_like(q)
    lib.ora_compute_flux_edge(0, ni, oracle.ptr(edges), oracle.ptr(q), oracle.ptr(f))
    lib.ora_compute_boundary_flux_edge(ni, nb, oracle.ptr(edges), oracle.ptr(q), oracle.ptr(f))
    lib.ora_compute_wall_flux_edge(ni + nb, nw, oracle.ptr(edges), oracle.ptr(q), oracle.ptr(f), C.byref(ff))
    _bits_equal(s.get(0, "fluxes"), f, "all three flux loops at size on the mixed level")
    s.zero_fluxes(0)
    v, sf = q.copy(), np.zeros(nel)
    f[:] = 0.0
    for sweep in range(2):
        old = v.copy()
        lib.ora_compute_step_factor(nel, oracle.ptr(v), oracle.ptr(vol), oracle.ptr(sf))
        for j in range(3):
            lib.ora_compute_flux_edge(0, ni, oracle.ptr(edges), oracle.ptr(v), oracle.ptr(f))
            lib.ora_compute_boundary_flux_edge(ni, nb, oracle.ptr(edges), oracle.ptr(v), oracle.ptr(f))
            lib.ora_compute_wall_flux_edge(ni + nb, nw, oracle.ptr(edges), oracle.ptr(v), oracle.ptr(f), C.byref(ff))
            lib.ora_time_step(j, nel, oracle.ptr(sf), oracle.ptr(f), oracle.ptr(old), oracle.ptr(v))
        s.smooth(0, 1)
        _bits_equal(s.get(0, "variables"), v, f"sweep {sweep}: variables")
        _bits_equal(s.get(0, "residuals"), v - old, f"sweep {sweep}: residuals")
    s.close()


def test_cfg1_fvcorr_like_97k_nodes_1000_iterations(oracle):
    """BASELINE configs[0] in its synthetic form at its full 1,000 iterations: mesh_name = fvcorr,
    46^3 box minus its centre node = 97,335 nodes, undamped weights, local time step; state bit for bit, RMS history to
    1e-12, then the reference's own -v rule.  (The oracle itself is pinned against the REFERENCE BINARY on this very
    input by tests/test_host_configs.py.)"""
    import mgcfd
    from mgcfd import meshgen
    mg = meshgen.make_multigrid((46,), "fvcorr", seed=0, cavity_radius=0.001)
    levels = mgcfd.generated_to_levels(mg)
    assert levels[0]["nel"] == 97335
    iters = 1000
    lv, keep = _oracle_levels(oracle, levels)
    lib = oracle.load()
    want_rms = np.zeros(iters)
    assert lib.ora_solve(lv, 1, 0, iters, 0, oracle.ptr(want_rms), None) == 0
    s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    rms = s.run_cycles(iters)
    got = s.get(0, "variables")
    assert s.loop_iters(0)["flux"] == 3 * iters * levels[0]["n_internal"]
    s.close()
    _bits_equal(got, keep[0]["variables"], "fvcorr-like 97K nodes, 1,000 iterations")
    assert np.allclose(rms, want_rms, rtol=1e-12, atol=0)
    assert lib.ora_identify_differences(oracle.ptr(np.ascontiguousarray(got)), oracle.ptr(keep[0]["variables"]), levels[0]["nel"], 0) == -1


def _in_process_ranks(n, body):
    """Run body(rank) on n threads (ranks of one process sharing this GPU); re-raise the first error."""
    errors = []

    def run(r):
        try:
            body(r)
        except Exception as e:                           # pragma: no cover
            errors.append(e)
    threads = [threading.Thread(target=run, args=(r,)) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_cfg5_eight_way_partition_of_the_8x_level(oracle):
    """BASELINE configs[4] at size on one GPU: the 134^3 connected level (2,406,104 nodes / 7,164,444 edges) split by
    recursive coordinate bisection into 8 parts = 8 solvers here (threads for ranks, an in-process copy where RCCL
    sends), one fused partitioned sweep (stage launch + halo message, three times; global-min time step over the
    parts) — owned nodes of every part against the ORACLE's sweep of the whole mesh, bit for bit."""
    import torch
    import bench
    import mgcfd
    from mgcfd.distributed import HipSolverAdapter, PartitionedSweep
    from mgcfd.partition import partition_level, rcb_partition
    n_parts = 8
    mg, levels = bench.build_workload(bench.LATTICE_8X)
    L = levels[0]
    assert L["nel"] == 134 ** 3
    nel, ni, nb, nw = L["nel"], L["n_internal"], L["n_boundary"], L["n_wall"]
    parts = partition_level(L, rcb_partition(np.asarray(L["coords"]), n_parts))
    assert sum(p.n_owned for p in parts) == nel
    # the oracle's sweep of the whole level
    lib = oracle.load()
    edges = np.ascontiguousarray(L["edges"]).copy()
    coords = np.ascontiguousarray(L["coords"], dtype=np.float64)
    lib.ora_adjust_ewt(oracle.ptr(coords), len(edges), oracle.ptr(edges))
    lib.ora_dampen_ewt(len(edges), oracle.ptr(edges), 5e-8)
    ff = oracle.farfield()
    vol = np.ascontiguousarray(L["volumes"], dtype=np.float64)
    q0 = bench.perturbed_state(nel, np.array(ff.var))
    v, old, f, sf = q0.copy(), q0.copy(), np.zeros_like(q0), np.zeros(nel)
    lib.ora_compute_step_factor(nel, oracle.ptr(v), oracle.ptr(vol), oracle.ptr(sf))
    for j in range(3):
        lib.ora_compute_flux_edge(0, ni, oracle.ptr(edges), oracle.ptr(v), oracle.ptr(f))
        lib.ora_compute_boundary_flux_edge(ni, nb, oracle.ptr(edges), oracle.ptr(v), oracle.ptr(f))
        lib.ora_compute_wall_flux_edge(ni + nb, nw, oracle.ptr(edges), oracle.ptr(v), oracle.ptr(f), C.byref(ff))
        lib.ora_time_step(j, nel, oracle.ptr(sf), oracle.ptr(f), oracle.ptr(old), oracle.ptr(v))
    del edges, coords, f

    dev = torch.device("cuda", 0)
    tstream = torch.cuda.Stream()
    solvers, sweepers = [], []
    barrier = threading.Barrier(n_parts)

    def exchange(sw):
        barrier.wait()                                   # every part has enqueued its packs
        for peer, buf in sw.buf_recv.items():
            buf.copy_(sweepers[peer].buf_send[sw.part.rank])
        barrier.wait()                                   # nobody repacks before all copies are enqueued

    def allreduce_min(sw, level=0, partials=False):
        barrier.wait()
        if sw.part.rank == 0:
            m = torch.stack([x.s.min_tensor(0) for x in sweepers]).min(dim=0).values
            for x in sweepers:
                x.s.min_tensor(0).copy_(m)
        barrier.wait()

    for P in parts:
        s = mgcfd.Solver.from_arrays([P.level], mg.mesh_variant, n_owned=[P.n_owned])
        s.set_stream(tstream.cuda_stream)
        s.set(0, "variables", q0[P.global_ids])          # ghosts start current
        solvers.append(s)
        sweepers.append(PartitionedSweep(HipSolverAdapter(s, dev), P, None, exchange=exchange, allreduce_min=allreduce_min,
                                         make_buffer=lambda n: torch.empty(max(n, 1), dtype=torch.float64, device=dev), fused=True))

    def body(r):
        torch.cuda.set_device(0)
        torch.cuda.set_stream(tstream)                   # the current stream is per thread
        sweepers[r].sweep()
    _in_process_ranks(n_parts, body)
    for P, s in zip(parts, solvers):
        own = P.global_ids[:P.n_owned]
        _bits_equal(s.get(0, "variables")[:P.n_owned], v[own], f"part {P.rank}: owned variables")
        _bits_equal(s.get(0, "variables")[P.n_owned:], v[P.global_ids[P.n_owned:]], f"part {P.rank}: ghosts after the last message")
        _bits_equal(s.get(0, "step_factors")[:P.n_owned], sf[own], f"part {P.rank}: step factors")
        s.close()


def test_cfg4_one_level_per_solver_at_full_size(oracle):
    """BASELINE configs[3] (one multigrid level per GPU) at size, in process: four solvers, solver l sweeps only level
    l, the restricted variables / coarse residuals move between them as whole-array device copies (where RCCL
    point-to-point messages go); 2 V-cycles against ora_solve on every level's owner."""
    import torch
    import bench
    import mgcfd
    from mgcfd.distributed import HipSolverAdapter, LevelPerRankCycle
    mg, levels = bench.build_hierarchy()
    n = len(levels)
    cycles = 2
    lv, keep = _oracle_levels(oracle, levels)
    lib = oracle.load()
    want_rms = np.zeros(cycles)
    assert lib.ora_solve(lv, n, mg.mesh_variant, cycles, 0, oracle.ptr(want_rms), None) == 0
    dev = torch.device("cuda", 0)
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    solvers = []
    for r in range(n):
        s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
        s.set_stream(tstream.cuda_stream)
        solvers.append(HipSolverAdapter(s, dev))
    mailbox = {}
    lock = threading.Condition()

    def send(r):
        def f(t, dst):
            with lock:
                mailbox[(r, dst)] = t
                lock.notify_all()
        return f

    def recv(r):
        def f(t, src):
            with lock:
                lock.wait_for(lambda: (src, r) in mailbox, timeout=120)
                t.copy_(mailbox.pop((src, r)))
        return f

    def body(r):
        torch.cuda.set_device(0)
        torch.cuda.set_stream(tstream)
        cyc = LevelPerRankCycle(solvers[r], n, r, n, send=send(r), recv=recv(r))
        for _ in range(cycles):
            cyc.cycle()
        torch.cuda.synchronize()
    _in_process_ranks(n, body)
    for l in range(n):
        _bits_equal(solvers[l].get(l, "variables"), keep[l]["variables"], f"level {l} on its owner")
    for s in solvers:
        s.s.close()
    torch.cuda.set_stream(torch.cuda.default_stream())


def test_bench_line_at_the_drivers_arguments():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` — what the driver runs — prints ONE JSON line with the contract's
    fields: BASELINE's metric and unit, whole-job value, steps / warmup echoed, the roofline object of the flux kernel
    (HBM bound, achieved / peak / frac consistent, traffic named with the profile it came from), the V-cycle leg and the
    CPU baseline of the same level (bounded sample)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (the timed region of this command is 1.1 ms: one pause of the host inside it — it has been seen once, 109 us per step —
    #  halves the figure.  The figure's lower bound below is there to catch a sweep that silently left the fused path, so a
    #  run below it is repeated once before it counts.)
    for attempt in range(2):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "2"],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        d = json.loads(lines[0])
        if d["value"] > 30000:
            break
    assert d["metric"].startswith("Medges/s (compute_flux_edge)") and d["unit"] == "Medges/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert 30000 < d["value"] < 80000 and abs(d["value"] - 3 * 888822 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-3 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["avg_kernel_us"] * 1e-6) / 1e9) < 1.0
    assert roof["algorithmic_bytes_per_launch"] == 40 * 888822 + 80 * 300763
    assert 0.3 < roof["frac"] < 0.8 and 10.0 < roof["empirical_ceiling_us"] < roof["avg_kernel_us"]
    assert roof["traffic"] is None or (roof["traffic"] >= roof["algorithmic_bytes_per_launch"] and "profiles/" in roof["traffic_source"])
    assert roof["fused_stage"]["launches"] == 60
    # a fused stage priced on what it must move (40 E + 128 N) beside the two-loop figure, and the traffic ratios as numbers
    fs = roof["fused_stage"]
    assert fs["bytes_a_fused_stage_must_move"] == 40 * 888822 + 128 * 300763 and 0.3 < fs["frac_of_bytes_a_fused_stage_must_move"] < fs["frac"]
    assert roof["traffic"] is None or abs(roof["traffic_over_algorithmic"] - roof["traffic"] / roof["algorithmic_bytes_per_launch"]) < 1e-3
    # the practical ceiling: a tile-shaped stream of exactly the algorithmic bytes, faster than the kernel and slower than the chip's peak
    assert 7.45 < roof["practical_ceiling_us"] < roof["avg_kernel_us"] and abs(roof["flux_over_practical_ceiling"] - roof["avg_kernel_us"] / roof["practical_ceiling_us"]) < 2e-3
    # two independent batches of the launch in flight at once: what the kernel reaches when the chip is kept full — beside the
    # line's figure (one launch after the other), never as it; faster per launch than one batch alone, and within the peak
    two = roof["two_launches_in_flight"]
    assert "error" not in two, two
    assert two["sweeps"]["us_per_sweep_two_at_once"] < two["sweeps"]["us_per_sweep_one_solver"] and two["sweeps"]["gain"] > 1.05, two["sweeps"]
    for name in ("bit_identical", "order_free"):
        assert 7.45 < two[name]["us_per_launch"] < roof["avg_kernel_us"] and abs(two[name]["frac"] - roof["algorithmic_bytes_per_launch"] / (two[name]["us_per_launch"] * 1e-6) / 8e12) < 2e-3, two
    # (FMA contraction allowed: the same launch, a little faster, reported beside the bit-identical figure)
    assert 10.0 < roof["fma_contracted"]["avg_kernel_us"] < 1.05 * roof["avg_kernel_us"] and "1e-12" in roof["fma_contracted"]["numerics"]
    assert 0.0002 < d["vcycle"]["wall_s_per_cycle"] < 0.001
    cpu = d["cpu_baseline"]
    assert cpu["unit"] == "Medges/s" and cpu["cores"] == 1 and cpu["kind"] in ("reference", "port") and cpu["value"] > 1 and cpu["sample"] and cpu["cpu_model"]
    # (the threaded figure says how many threads it ran and how many cores the process may use)
    mt = cpu["multi_thread"]
    assert 1 <= mt["threads"] <= mt["cores_usable_by_this_process"] and mt["value"] > cpu["value"]


def test_bench_rehearsal_two_ranks_on_this_gpu():
    """`python bench.py --gpus 2` from a plain invocation starts two ranks itself (here both on device 0 with gloo
    collectives: MGCFD_BENCH_REHEARSAL=1, a functional rehearsal of the N > 1 path, not a measurement) and reports
    n_gpus = 2 from the process group; partitioned (the N > 1 default) and mesh copies."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MGCFD_BENCH_REHEARSAL="1")
    for extra, kind, scaling in ((["--lattice", "30", "--vcycle-hierarchy", "tiny"], "partitioned", "strong"), (["--workload", "copies", "--lattice", "24"], "copies", "weak"),
                                 (["--lattice", "30", "--exchange", "ipc", "--no-vcycle"], "partitioned", "strong")):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"] + extra,
                           capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["config"]["workload_kind"] == kind and line["scaling"] == scaling
        assert line["value"] > 0
        if kind == "partitioned":                           # (--exchange auto tries the IPC form first; explicit or not, it must have run)
            # the LIBRARY's rank loop ran (two processes storing into each other through HIP IPC, the time-step all-reduce through
            # the same flags) and had reproduced the torch path's sweep bit for bit at start-up — no fall-back
            assert "HIP IPC" in line["config"]["exchange"] and "checked against the torch path" in line["config"]["exchange"]
            assert "not used" not in line["config"]["exchange"] and "bit for bit on every rank" in line["config"]["exchange"]
            if "--no-vcycle" not in extra:
                # ... and the V-cycle half of the metric on the partitioned hierarchy (the torch orchestration here: RCCL cannot
                # form a communicator with both ranks on one device; on N GPUs the library's mgcfd_rank_cycles)
                vc = line["vcycle"]
                assert vc["wall_s_per_cycle"] > 0 and vc["state_valid"] and "every level partitioned over 2 ranks" in vc["workload"]
                assert "MG V-cycle wall-s" in line["metric"]


@pytest.mark.parametrize("fail,expect", [("attach", "the library's rank loop not used"), ("phase", "not used"),
                                         ("ipc-start", "differs from the torch path"), ("ipc-end", "discarded after the run"),
                                         ("vcycle-setup", "the library's cycle loop not used"), ("vcycle-cycles", "mgcfd_rank_cycles not used"),
                                         ("group-setup", "injected failure (group-setup)"), ("group-end", "counts"),
                                         ("rank-graphs-rehearsal", "Error"), ("rank-graphs-rehearsal,rank-graphs-setup", "going back to the buffered RCCL form failed on some rank")])
def test_bench_form_ladder_survives_injected_failures(fail, expect):
    """What the first run on several GPUs will execute, with something breaking on ONE rank at every rung of bench.py's ladder
    (MGCFD_BENCH_FAIL, rehearsal: both ranks on this device over gloo): the library's rank set-up raising, its start-up sweep
    raising mid-phase, the IPC form failing its start-up check, the IPC form's final state failing its check; in the V-cycle leg
    the library's set-up raising and a rank finding something wrong just before mgcfd_rank_cycles (every rank must then stay out
    of that loop: round 3's advisor finding); in the in-process group leg its set-up raising and its final state failing the
    check; the rank-graphs leg entered without an RCCL form, and with its set-up raising on one rank.  Every time the run must END — within the bound, never a stuck rank — with a valid line from a fall-back."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MGCFD_BENCH_REHEARSAL="1", MGCFD_BENCH_FAIL=fail, MGCFD_BENCH_WATCHDOG_S="240", MGCFD_BENCH_LEG_S="200")
    legs = ["--vcycle-hierarchy", "tiny", "--no-group"] if fail.startswith("vcycle") else (["--no-vcycle"] if fail.startswith("group") else ["--no-vcycle", "--no-group"])
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--lattice", "30"] + legs,
                       capture_output=True, text=True, env=env, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["workload_kind"] == "partitioned"
    ex = line["config"]["exchange"]
    if fail.startswith("vcycle"):
        # the V-cycle leg ended on the torch orchestration, said why, and nobody was left inside the library's loop
        vc = line["vcycle"]
        assert any(expect in n for n in vc["notes"]), vc["notes"]
        assert vc["wall_s_per_cycle"] > 0 and vc["state_valid"] and vc["form"].startswith("torch.distributed")
        return
    if fail.startswith("rank-graphs"):
        # the leg that replays the RCCL form's sweeps from hipGraphs, entered although this rehearsal has no RCCL form (both ranks on
        # one device): it must end with an error record on every rank at the same place and leave the line's figure alone
        assert expect in line["rank_graphs"]["error"], line["rank_graphs"]
        assert "us_per_sweep" not in line["rank_graphs"]
        return
    if fail.startswith("group"):
        g = line["in_process_group"]
        if fail == "group-setup":
            assert expect in g["error"], g
        else:
            assert g["sweeps"]["counts"] is False and g["sweeps"]["nodes_differing_from_the_unpartitioned_level"] == 1, g
        return
    assert expect in ex, ex
    # what the library itself says it is a rank of (not what the launcher said)
    assert line["config"]["halo_bytes_per_stage_rank0_sent"] > 0
    if fail == "attach":
        assert ex.startswith("torch.distributed")                           # nothing of the library's loop was usable: the torch path ran
    assert line["config"]["library_ranks"]["ranks"] == 2                    # (rank 0's own attachment succeeded in every case)


def test_bench_prints_the_torch_paths_line_when_a_library_form_never_comes_back():
    """The library's rank loops have run on ONE GPU only, and a rank stuck inside a collective cannot be interrupted: bench.py
    measures the torch path over the same W + K sweeps FIRST and arms a guard around the library's rungs; if they have not come back
    in time (here: one rank sleeps for ever inside the start-up sweep, the guard at 40 s) rank 0 prints the torch path's line — a
    valid line, saying what happened — and every rank leaves."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MGCFD_BENCH_REHEARSAL="1", MGCFD_BENCH_FAIL="hang", MGCFD_BENCH_LADDER_S="40", MGCFD_BENCH_WATCHDOG_S="300")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--lattice", "30", "--no-vcycle", "--no-group"],
                       capture_output=True, text=True, env=env, timeout=400)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stderr[-2000:]
    line = json.loads(lines[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["steps"] == 5 and line["config"]["workload_kind"] == "partitioned"
    assert "did not come back" in line["config"]["exchange"] and "did not end within" in line["library_forms"]["error"]


def test_bench_in_process_group_leg_on_this_gpu():
    """`bench.py --gpus 2` (rehearsal: both ranks and both group members on this device): beside the ranks' figure the line
    carries `in_process_group` — the level and the hierarchy swept by ONE process through mgcfd_group_sweeps / mgcfd_group_cycles —
    and its sweeps' final state equals the unpartitioned level's bit for bit (a figure that fails that check says so)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MGCFD_BENCH_REHEARSAL="1", MGCFD_BENCH_WATCHDOG_S="300")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--lattice", "30", "--vcycle-hierarchy", "tiny"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    g = line["in_process_group"]
    assert "error" not in g, g
    assert g["ranks"] == 2 and "REHEARSAL" in g["form"]
    assert g["sweeps"]["counts"] is True and g["sweeps"]["nodes_differing_from_the_unpartitioned_level"] == 0 and g["sweeps"]["value"] > 0
    assert g["vcycle"]["wall_s_per_cycle"] > 0 and g["vcycle"]["state_valid"]
    assert line["vcycle"]["wall_s_per_cycle"] > 0                              # (the ranks' own V-cycle leg beside it)


def _group_sweeps_check(mg, n_parts, sweeps, partitioner="rcb"):
    """Level 0 of `mg` split into n_parts solvers, the sweeps run by the LIBRARY's own loop (mgcfd_group_sweeps: boundary
    tiles, one pack, device-to-device messages, interior tiles meanwhile, one unpack) against the unpartitioned run."""
    import mgcfd
    from conftest import perturbed_state
    from mgcfd.partition import partition_level, rcb_partition, slab_partition
    L = mgcfd.generated_to_levels(mg)[0]
    parts = partition_level(L, (slab_partition if partitioner == "slab" else rcb_partition)(np.asarray(L["coords"]), n_parts))
    whole = mgcfd.Solver.from_arrays([L], mg.mesh_variant)
    q0 = perturbed_state(L["nel"], whole.far_field()[:5], seed=21)
    whole.set(0, "variables", q0)
    whole.smooth(0, sweeps)
    want_v, want_res, want_sf, want_rms = whole.get(0, "variables"), whole.get(0, "residuals"), whole.get(0, "step_factors"), whole.calc_rms(0)
    whole.close()
    solvers = []
    for P in parts:
        s = mgcfd.Solver.from_arrays([P.level], mg.mesh_variant, n_owned=[P.n_owned])
        s.set(0, "variables", q0[P.global_ids])
        solvers.append(s)
    g = mgcfd.Group(solvers)
    for P, s in zip(parts, solvers):
        s.rank_set_halo(0, P)
        info = s.rank_halo_info(0)
        assert info["nodes_sent"] == sum(len(v) for v in P.send.values()) and info["nodes_received"] == sum(len(v) for v in P.recv.values())
        assert info["boundary_tiles"] + info["interior_tiles"] == -(-P.n_owned // 256)     # (the owned nodes are numbered first: tiles of ghosts only are never launched)
    g.exchange(0)
    g.sweeps(0, sweeps)
    g.synchronize()
    for P, s in zip(parts, solvers):
        own = P.global_ids[:P.n_owned]
        _bits_equal(s.get(0, "variables")[:P.n_owned], want_v[own], f"part {P.rank}: owned variables")
        _bits_equal(s.get(0, "variables")[P.n_owned:], want_v[P.global_ids[P.n_owned:]], f"part {P.rank}: ghosts")
        _bits_equal(s.get(0, "residuals")[:P.n_owned], want_res[own], f"part {P.rank}: residuals")
        _bits_equal(s.get(0, "step_factors")[:P.n_owned], want_sf[own], f"part {P.rank}: step factors")
    got_rms = g.rms(0)
    # (a random level may blow up — both sides then hold the same non-finite state, bit for bit above, and a NaN RMS)
    assert (np.isnan(got_rms) and np.isnan(want_rms)) or got_rms == want_rms or abs(got_rms - want_rms) <= 1e-12 * abs(want_rms), (got_rms, want_rms)
    # ... and again from the start with the RMS of every sweep gathered on the devices (mgcfd_group_sweeps_rms: what the
    # drop-in's --gpus N loop calls), against calc_rms of the whole level after each sweep
    whole = mgcfd.Solver.from_arrays([L], mg.mesh_variant)
    whole.set(0, "variables", q0)
    want_each = []
    for _ in range(sweeps):
        whole.smooth(0, 1)
        want_each.append(whole.calc_rms(0))
    whole.close()
    for P, s in zip(parts, solvers):
        s.set(0, "variables", q0[P.global_ids])
    g.exchange(0)
    got_each = g.sweeps_rms(0, sweeps)
    assert np.allclose(got_each, want_each, rtol=1e-12, atol=0.0, equal_nan=True), (got_each, want_each)
    for P, s in zip(parts, solvers):
        _bits_equal(s.get(0, "variables")[:P.n_owned], want_v[P.global_ids[:P.n_owned]], f"part {P.rank}: owned variables (second run)")
        _bits_equal(s.get(0, "variables")[P.n_owned:], want_v[P.global_ids[P.n_owned:]], f"part {P.rank}: ghosts (second run)")
    g.close()
    for s in solvers:
        s.close()


@pytest.mark.parametrize("kind,n_parts,sweeps", [("lattice", 2, 3), ("lattice", 5, 3), ("fvcorr", 3, 3), ("tet", 4, 3),
                                                 ("lattice", 5, 7), ("fvcorr", 3, 6), ("tet", 4, 5)])
def test_library_runs_the_partitioned_sweeps_itself(kind, n_parts, sweeps):
    """mgcfd_group_sweeps (the C++ host's own loop over the ranks of one process) on small levels: a lattice, a
    local-time-step (fvcorr) level, a tetrahedral level with long rows — owned nodes and ghosts bit for bit against
    mgcfd_smooth on the whole level, RMS over the owned nodes of all parts.  Calls of four sweeps or more are issued by
    a host thread per rank (a barrier per stage between them), shorter ones by the calling thread alone."""
    from mgcfd import meshgen
    if kind == "tet":
        mg = meshgen.make_tet_multigrid((5000,), "m6wing", seed=6)
    elif kind == "fvcorr":
        mg = meshgen.make_multigrid((14,), "fvcorr", seed=4, cavity_radius=0.01, volume_noise=0.02)
    else:
        mg = meshgen.make_multigrid((20,), "m6wing", seed=4, cavity_radius=0.15, jitter=0.2, area_noise=0.05, volume_noise=0.05)
    _group_sweeps_check(mg, n_parts, sweeps)


@pytest.mark.parametrize("form", ["buffered", "one_thread"])
def test_library_loop_in_its_other_forms(form, monkeypatch):
    """The forms the group loop falls back to or can be switched to: messages through buffers, a second stream and
    pack / copy / unpack (MGCFD_GROUP_DIRECT=0: more than eight peers, or a plan that mixes ghosts into the tiles), and one
    host thread issuing every rank's launches (MGCFD_GROUP_THREADS=0) — same results, bit for bit."""
    from mgcfd import meshgen
    monkeypatch.setenv("MGCFD_GROUP_DIRECT" if form == "buffered" else "MGCFD_GROUP_THREADS", "0")
    mg = meshgen.make_multigrid((20,), "m6wing", seed=4, cavity_radius=0.15, jitter=0.2, area_noise=0.05, volume_noise=0.05)
    _group_sweeps_check(mg, 4, 6)
    mg = meshgen.make_multigrid((14,), "fvcorr", seed=4, cavity_radius=0.01, volume_noise=0.02)
    _group_sweeps_check(mg, 3, 5)


@pytest.mark.parametrize("ranks,lattice,sweeps,mesh,extra", [(2, 14, 5, "fvcorr", []), (3, 20, 7, "fvcorr", []), (2, 16, 6, "m6wing", []), (4, 22, 9, "m6wing", []),
                                                             (3, 20, 7, "m6wing", ["--unsplit"]), (2, 30, 8, "tet", ["--one-by-one"]),
                                                             (2, 16, 6, "m6wing", ["--fused"]), (4, 24, 9, "m6wing", ["--fused"]), (3, 20, 7, "fvcorr", ["--fused"]),
                                                             (3, 40, 5, "tet", ["--fused", "--one-by-one"])])
def test_ranks_in_different_processes_store_into_each_other_through_hip_ipc(ranks, lattice, sweeps, mesh, extra):
    """tools/ipc_ranks_check.py: `ranks` PROCESSES on this one GPU, each with its part of a local-time-step level, the state
    buffers and flag words of its neighbours opened through HIP IPC (mgcfd_rank_ipc_export / _attach): a stage's message is one
    launch that stores into the neighbours' ghost slots and raises their flags, the next stage waits for them.  Every rank
    compares its owned nodes and its ghosts with the unpartitioned level, bit for bit, and no wait may have given up.  With a
    global time step (m6wing) every rank also stores its minimum into every other rank's memory each sweep: the all-reduce
    without a collective library."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "ipc_ranks_check.py"), "--ranks", str(ranks), "--lattice", str(lattice), "--sweeps", str(sweeps), "--mesh", mesh] + extra,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("owned equal, ghosts equal, waits that gave up: 0") == ranks


def test_torch_path_over_gloo_on_this_gpu_equals_the_whole_level():
    """The torch.distributed path of a partitioned level as bench.py's one-GPU rehearsal runs it — two processes, gloo, the
    messages in DEVICE tensors (tools/torch_path_check.py): seven sweeps must equal the whole level on every rank.  (gloo's
    point-to-point calls know nothing of streams; mgcfd/distributed.py stages such messages through the host.)  It is the
    reference bench.py checks the library's rank loops against, so it has to be right itself."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29731",
                        os.path.join(root, "tools", "torch_path_check.py"), "7"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count(": 0 node(s) differ from the whole level") == 2, r.stdout[-1000:]


def test_cfg5_eight_parts_library_loop():
    """BASELINE configs[4] at size with the sweep loop in the library: the 134^3 level in 8 parts as an in-process group
    (all on this GPU), four sweeps — a host thread per rank —, against mgcfd_smooth on the whole 2.4 M-node level."""
    import bench
    mg, _ = bench.build_workload(bench.LATTICE_8X)
    _group_sweeps_check(mg, 8, 4)


def test_rccl_loads_and_a_one_rank_communicator_sweeps():
    """The RCCL transport with the one rank this box has: librccl is found at run time, ncclCommInitRank succeeds, and
    mgcfd_rank_sweeps (all-reduce on one rank, no peers) equals mgcfd_smooth bit for bit."""
    import mgcfd
    from conftest import perturbed_state
    from mgcfd import meshgen
    from mgcfd.partition import partition_level, rcb_partition
    mg = meshgen.make_multigrid((16,), "m6wing", seed=2, jitter=0.2, area_noise=0.05, volume_noise=0.05)
    L = mgcfd.generated_to_levels(mg)[0]
    P = partition_level(L, rcb_partition(np.asarray(L["coords"]), 1))[0]
    ref = mgcfd.Solver.from_arrays([L], mg.mesh_variant)
    q0 = perturbed_state(L["nel"], ref.far_field()[:5], seed=3)
    ref.set(0, "variables", q0)
    ref.smooth(0, 3)
    want = ref.get(0, "variables")
    ref.close()
    s = mgcfd.Solver.from_arrays([P.level], mg.mesh_variant, n_owned=[P.n_owned])
    s.set(0, "variables", q0[P.global_ids])
    s.rank_attach_rccl(0, 1, mgcfd.rccl_unique_id())
    s.rank_set_halo(0, P)
    s.rank_exchange(0)
    s.rank_sweeps(0, 3)
    _bits_equal(s.get(0, "variables"), want, "one-rank RCCL sweeps")
    assert s.rank_residual_sumsq(0) > 0
    # ... and replayed from captured hipGraphs (MGCFD_OPT_GRAPH): seven sweeps pass through all three buffer rotations twice
    ref = mgcfd.Solver.from_arrays([L], mg.mesh_variant)
    ref.set(0, "variables", q0)
    ref.smooth(0, 7)
    s.set(0, "variables", q0[P.global_ids])
    s.set_option("graph", 1)
    s.rank_exchange(0)
    s.rank_sweeps(0, 7)
    _bits_equal(s.get(0, "variables"), ref.get(0, "variables"), "one-rank RCCL sweeps replayed from hipGraphs")
    # (what bench.py's rank_graphs leg asks before it believes a replayed figure: graphs exist, none refused, sweeps really replayed)
    st = s.rank_graph_status(0)
    assert st["graphs"] == 3 and not st["capture_refused"] and st["sweeps_replayed"] >= 4, st
    assert s.loop_iters(0)["flux"] == ref.loop_iters(0)["flux"] + 3 * 3 * L["n_internal"]
    ref.close()
    s.rank_detach()
    s.close()
