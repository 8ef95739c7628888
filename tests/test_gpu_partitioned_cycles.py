"""V-cycles on a PARTITIONED hierarchy swept inside the library (mgcfd_group_cycles; mgcfd_rank_cycles over RCCL with the one
rank a one-GPU box offers): every level of the hierarchy split over N solvers on this GPU — flux ghosts, the children a
rank's coarse nodes need for mg_restrict, the parents the prolongation reads — against mgcfd_run_cycles on the whole
hierarchy (itself bit-identical to the oracle and to the reference binary: tests/test_gpu_parity.py, test_gpu_configs.py):
bit for bit on the owned nodes of EVERY level, the RMS of every cycle to 1e-12 (sums over ranks are associated differently),
and the loop counters."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits_equal(got, want, what):
    got, want = np.ascontiguousarray(got), np.ascontiguousarray(want)
    bad = np.flatnonzero(got.ravel().view(np.int64) != want.ravel().view(np.int64))
    assert bad.size == 0, f"{what}: {bad.size} values differ bitwise; first {bad[:5]}, max abs diff {np.abs(got - want).max():.3e}"


def _group_over(levels, mesh_variant, n_parts, direct=True):
    import mgcfd
    from mgcfd.partition import partition_hierarchy, rcb_partition
    H = partition_hierarchy(levels, rcb_partition(np.asarray(levels[0]["coords"]), n_parts))
    assert all(sum(h.levels[l].n_owned for h in H) == levels[l]["nel"] for l in range(len(levels)))
    solvers = []
    for h in H:
        lv, owned, keys = h.solver_args()
        solvers.append(mgcfd.Solver.from_arrays(lv, mesh_variant, n_owned=owned, order_keys=keys))
    g = mgcfd.Group(solvers)
    for h, s in zip(H, solvers):
        for l in range(len(levels)):
            s.rank_set_halo(l, h.levels[l])
    for l in range(len(levels)):
        g.exchange(l)
    return H, solvers, g


def _check_against_whole(levels, mesh_variant, n_parts, cycles, exact=True, whole_variant=-1, rel=0.0):
    import mgcfd
    whole = mgcfd.Solver.from_arrays(levels, mesh_variant)
    whole.set_option("exact", int(exact))
    whole.set_option("flux_variant", whole_variant)
    want_rms = whole.run_cycles(cycles)
    want = [(whole.get(l, "variables"), whole.get(l, "residuals")) for l in range(len(levels))]
    want_iters = [whole.loop_iters(l) for l in range(len(levels))]
    whole.close()
    H, solvers, g = _group_over(levels, mesh_variant, n_parts)
    for s in solvers:
        s.set_option("exact", int(exact))
    rms = g.cycles(cycles)
    for h, s in zip(H, solvers):
        for l in range(len(levels)):
            P = h.levels[l]
            own = P.global_ids[:P.n_owned]
            if rel > 0.0:
                for k, nm in ((0, "variables"), (1, "residuals")):
                    got, ref = s.get(l, nm)[:P.n_owned], want[l][k][own]
                    scale = np.abs(want[l][0]).max()              # (residuals are differences of the variables: same scale)
                    assert np.abs(got - ref).max() <= rel * scale, f"{n_parts} parts, rank {h.rank}, level {l}: owned {nm} differ by {np.abs(got - ref).max() / scale:.2e} relative"
                continue
            _bits_equal(s.get(l, "variables")[:P.n_owned], want[l][0][own], f"{n_parts} parts, rank {h.rank}, level {l}: owned variables")
            _bits_equal(s.get(l, "residuals")[:P.n_owned], want[l][1][own], f"{n_parts} parts, rank {h.rank}, level {l}: owned residuals")
    assert np.allclose(rms, want_rms, rtol=1e-12, atol=0.0), (rms, want_rms)
    # LoopNumIters: a rank's counters hold its own share; node loops add up to the whole level's (edges cut by the partition
    # are walked once per side)
    for l in range(len(levels)):
        for k in ("compute_step", "time_step"):
            owned_share = sum(s.loop_iters(l)[k] * h.levels[l].n_owned // max(h.levels[l].level["nel"], 1) for h, s in zip(H, solvers))
            assert owned_share <= want_iters[l][k]
    g.close()
    for s in solvers:
        s.close()


@pytest.mark.parametrize("sizes,n_parts", [((12, 6, 3), 3), ((14, 7), 4), ((13, 9, 6), 2), ((16, 10, 6, 4), 5)])
def test_group_cycles_on_lattice_hierarchies_equal_the_whole(sizes, n_parts):
    from mgcfd import meshgen
    mg = meshgen.make_multigrid(sizes, "m6wing", seed=4, jitter=0.2, area_noise=0.05, volume_noise=0.05)
    import mgcfd
    _check_against_whole(mgcfd.generated_to_levels(mg), mg.mesh_variant, n_parts, cycles=3)


@pytest.mark.parametrize("threads", ["0", "1"])
def test_group_cycles_with_and_without_a_host_thread_per_rank(monkeypatch, threads):
    """MGCFD_GROUP_THREADS=1: a host thread per rank issues that rank's calls (the default when every rank has a device of
    its own; barriers keep event records ahead of the waits for them); 0: the caller's thread issues everything (the default
    when the ranks share one device, as here)."""
    import mgcfd
    from mgcfd import meshgen
    monkeypatch.setenv("MGCFD_GROUP_THREADS", threads)
    mg = meshgen.make_multigrid((13, 9, 6), "m6wing", seed=6, jitter=0.2, area_noise=0.05, volume_noise=0.05)
    _check_against_whole(mgcfd.generated_to_levels(mg), mg.mesh_variant, 4, cycles=3)
    if threads == "1":
        import bench
        mg, levels = bench.build_hierarchy()
        _check_against_whole(levels, mg.mesh_variant, 3, cycles=2)


def test_group_cycles_on_a_tetrahedral_hierarchy_and_with_the_local_time_step():
    import mgcfd
    from mgcfd import meshgen
    mg = meshgen.make_tet_multigrid((3000, 700, 150), "m6wing", seed=4)
    _check_against_whole(mgcfd.generated_to_levels(mg), mg.mesh_variant, 3, cycles=3)
    mg = meshgen.make_multigrid((12,), "fvcorr", seed=5, cavity_radius=0.01, volume_noise=0.02)       # single level, local time step
    _check_against_whole(mgcfd.generated_to_levels(mg), mg.mesh_variant, 3, cycles=4)


def test_group_cycles_with_contraction_allowed_stay_within_tolerance():
    """MGCFD_OPT_EXACT = 0 on every rank.  The partitioned cycles run the contracted NODE GATHER (a launch over part of a level,
    or one that must leave the ghost slots alone, never takes the order-free kernel): against the whole hierarchy swept by the same
    kernel (variant 1) they are the same operations, bit for bit; against the whole hierarchy in its automatic variant — since
    round 4 the order-free stages on lattice-like levels too, sums associated differently and not reproducible from run to run —
    within the fast mode's bound (tests/test_gpu_parity.py REL_FAST)."""
    import mgcfd
    from mgcfd import meshgen
    mg = meshgen.make_multigrid((12, 6, 3), "m6wing", seed=9, jitter=0.2, area_noise=0.05, volume_noise=0.05)
    _check_against_whole(mgcfd.generated_to_levels(mg), mg.mesh_variant, 3, cycles=2, exact=False, whole_variant=1)
    _check_against_whole(mgcfd.generated_to_levels(mg), mg.mesh_variant, 3, cycles=2, exact=False, rel=1e-12)


def test_group_cycles_at_full_size_in_three_and_eight_parts():
    """BASELINE configs[2]'s hierarchy (300,763 / 166,375 / 110,592 / 79,507 nodes) split over 3 and over 8 solvers on this
    one GPU: two V-cycles."""
    import bench
    mg, levels = bench.build_hierarchy()
    for n_parts in (3, 8):
        _check_against_whole(levels, mg.mesh_variant, n_parts, cycles=2)


def test_rank_cycles_over_rccl_with_one_rank():
    """The RCCL form with the one rank this box offers (no peers: the communicator, the all-reduces and the cycle's state
    machine run; the messages are empty): equals mgcfd_run_cycles."""
    import mgcfd
    from mgcfd import meshgen
    from mgcfd.partition import partition_hierarchy
    mg = meshgen.make_multigrid((12, 6, 3), "m6wing", seed=4, jitter=0.2, area_noise=0.05, volume_noise=0.05)
    levels = mgcfd.generated_to_levels(mg)
    whole = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    want_rms = whole.run_cycles(3)
    want = [whole.get(l, "variables") for l in range(len(levels))]
    whole.close()
    H = partition_hierarchy(levels, np.zeros(levels[0]["nel"], dtype=np.int64))
    lv, owned, keys = H[0].solver_args()
    s = mgcfd.Solver.from_arrays(lv, mg.mesh_variant, n_owned=owned, order_keys=keys)
    s.rank_attach_rccl(0, 1, mgcfd.rccl_unique_id())
    for l in range(len(levels)):
        s.rank_set_halo(l, H[0].levels[l])
        s.rank_exchange(l)
    rms = s.rank_cycles(3)
    for l in range(len(levels)):
        _bits_equal(s.get(l, "variables"), want[l][H[0].levels[l].global_ids], f"level {l}")
    assert np.allclose(rms, want_rms, rtol=1e-12, atol=0.0)
    s.rank_detach()
    s.close()
