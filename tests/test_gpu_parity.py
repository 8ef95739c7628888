"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bar: in EXACT mode every floating-point result is
bit-identical to the oracle (which is itself bit-identical to the reference built with
-ffp-contract=off); in FAST mode (FMA contraction allowed) values agree to 1e-12 relative.
Integer outputs (loop iteration counts) are always exact."""
import ctypes as C

import numpy as np
import pytest

from conftest import perturbed_state

pytestmark = pytest.mark.gpu

REL_FAST = 1e-12      # tolerance for MGCFD_OPT_EXACT = 0


def _assert_close(got, want, exact, what):
    if exact:
        bad = np.flatnonzero(got.ravel().view(np.int64) != want.ravel().view(np.int64))
        assert bad.size == 0, f"{what}: {bad.size} values differ bitwise; first {bad[:5]}, " \
                              f"max abs diff {np.abs(got - want).max():.3e}"
    else:
        scale = np.maximum(np.abs(want).max(), 1e-300)
        err = np.abs(got - want).max() / scale
        assert err <= REL_FAST, f"{what}: max relative error {err:.3e} > {REL_FAST}"


@pytest.fixture(scope="module")
def setup(mesh3_dir, oracle):
    import mgcfd
    mesh = mgcfd.Mesh("input.dat", mesh3_dir)
    solver = mgcfd.Solver.from_mesh(mesh)
    case = oracle.OracleCase.from_input_dat(mesh3_dir + "/input.dat")
    # oracle edge weights must get the same adjust/dampen the solver applied at creation
    lib = oracle.load()
    for l in range(case.nlevels):
        L = case.levels[l]
        lib.ora_adjust_ewt(L.coords, L.n_edges, L.edges)
        lib.ora_dampen_ewt(L.n_edges, L.edges, 5e-8)
    yield mgcfd, mesh, solver, case, lib
    solver.close()


@pytest.mark.parametrize("exact", [True, False])
def test_edge_weights_and_far_field(setup, oracle, exact):
    mgcfd, mesh, solver, case, lib = setup
    ff = oracle.farfield()
    want = np.concatenate([np.array(ff.var), np.array(ff.fc_mx), np.array(ff.fc_my), np.array(ff.fc_mz), np.array(ff.fc_de)])
    assert np.array_equal(solver.far_field(), want)
    for l in range(solver.num_levels):
        e = solver.get_edges(l, case.levels[l].n_edges)
        assert np.array_equal(e, case.edges(l)), f"level {l}: adjusted/dampened edge records differ"


@pytest.mark.parametrize("exact", [True, False])
def test_flux_kernels(setup, oracle, exact):
    mgcfd, mesh, solver, case, lib = setup
    solver.set_option("exact", int(exact))
    ff = oracle.farfield()
    for l in range(solver.num_levels):
        L = case.levels[l]
        q = perturbed_state(L.nel, ff.var, seed=100 + l)
        f0 = np.random.default_rng(7 + l).normal(size=(L.nel, 5))
        # each class on its own, accumulating onto a non-zero flux array
        steps = [("internal", lambda f: lib.ora_compute_flux_edge(L.internal_start, L.n_internal, L.edges, oracle.ptr(q), oracle.ptr(f)),
                  solver.compute_flux_edge),
                 ("boundary", lambda f: lib.ora_compute_boundary_flux_edge(L.boundary_start, L.n_boundary, L.edges, oracle.ptr(q), oracle.ptr(f)),
                  solver.compute_boundary_flux_edge),
                 ("wall", lambda f: lib.ora_compute_wall_flux_edge(L.wall_start, L.n_wall, L.edges, oracle.ptr(q), oracle.ptr(f), C.byref(ff)),
                  solver.compute_wall_flux_edge)]
        solver.set(l, "variables", q)
        want = f0.copy()
        solver.set(l, "fluxes", f0)
        for name, ora_fn, gpu_fn in steps:
            ora_fn(want)
            gpu_fn(l)
            _assert_close(solver.get(l, "fluxes"), want, exact, f"level {l} {name} flux")
        # fused launch from zero
        solver.zero_fluxes(l)
        solver.compute_fluxes(l)
        want = np.zeros((L.nel, 5))
        for _, ora_fn, _ in steps:
            ora_fn(want)
        _assert_close(solver.get(l, "fluxes"), want, exact, f"level {l} fused flux")
    solver.set_option("exact", 1)


@pytest.mark.parametrize("exact", [True, False])
def test_step_factor_and_time_step(setup, oracle, exact):
    mgcfd, mesh, solver, case, lib = setup
    solver.set_option("exact", int(exact))
    ff = oracle.farfield()
    l = 0
    L = case.levels[l]
    q = perturbed_state(L.nel, ff.var, seed=200)
    vol = case.array(l, "volumes")
    sf = np.zeros(L.nel)
    lib.ora_compute_step_factor(L.nel, oracle.ptr(q), oracle.ptr(vol), oracle.ptr(sf))
    solver.set(l, "variables", q)
    solver.compute_step_factor(l)
    _assert_close(solver.get(l, "step_factors"), sf, exact, "step factors")
    rng = np.random.default_rng(5)
    flux = rng.normal(size=(L.nel, 5)) * 1e-3
    old = q * (1.0 + 1e-3 * rng.uniform(-1, 1, q.shape))
    for j in range(3):
        want_f, want_v = flux.copy(), np.zeros_like(q)
        lib.ora_time_step(j, L.nel, oracle.ptr(sf), oracle.ptr(want_f), oracle.ptr(old), oracle.ptr(want_v))
        solver.set(l, "fluxes", flux)
        solver.set(l, "old_variables", old)
        solver.set(l, "step_factors", sf)
        solver.time_step(l, j)
        _assert_close(solver.get(l, "variables"), want_v, exact, f"time_step j={j} variables")
        assert not solver.get(l, "fluxes").any(), "time_step must zero the fluxes"
    solver.set_option("exact", 1)


@pytest.mark.parametrize("exact", [True, False])
def test_restrict_and_prolong(setup, oracle, exact):
    mgcfd, mesh, solver, case, lib = setup
    solver.set_option("exact", int(exact))
    ff = oracle.farfield()
    for l in range(solver.num_levels - 1):
        F, Cc = case.levels[l], case.levels[l + 1]
        qf = perturbed_state(F.nel, ff.var, seed=300 + l)
        qc = perturbed_state(Cc.nel, ff.var, seed=310 + l)
        want = qc.copy()
        scratch = np.zeros(Cc.nel, dtype=np.int64)
        lib.ora_mg_restrict(oracle.ptr(qf), oracle.ptr(want), Cc.nel, F.mg_map, oracle.ptr(scratch), F.mgc)
        solver.set(l, "variables", qf)
        solver.set(l + 1, "variables", qc)
        solver.restrict(l)
        _assert_close(solver.get(l + 1, "variables"), want, exact, f"restrict {l}->{l + 1}")

        rng = np.random.default_rng(320 + l)
        r_c = rng.normal(size=(Cc.nel, 5)) * 1e-4
        r_f = rng.normal(size=(F.nel, 5)) * 1e-4
        want_v = qf.copy()
        lib.ora_prolong_residuals_interpolate_proper(F.edges, F.n_internal, oracle.ptr(r_c), oracle.ptr(r_f),
                                                     oracle.ptr(want_v), F.nel, F.mg_map, Cc.coords, F.coords)
        solver.set(l, "variables", qf)
        solver.set(l, "residuals", r_f)
        solver.set(l + 1, "residuals", r_c)
        solver.prolong(l)
        _assert_close(solver.get(l, "variables"), want_v, exact, f"prolong {l + 1}->{l}")
    solver.set_option("exact", 1)


def _run_both(mgcfd, oracle, directory, cycles, exact, indirect_rw=False, duplicate=1, fuse=True, variant=0, graph=0):
    mesh = mgcfd.Mesh("input.dat", directory, duplicate)
    solver = mgcfd.Solver.from_mesh(mesh)
    solver.set_option("graph", graph)
    solver.set_option("exact", int(exact))
    solver.set_option("flux_variant", variant)
    solver.set_option("fuse_update", int(fuse))
    solver.set_option("indirect_rw", int(indirect_rw))
    rms = solver.run_cycles(cycles)
    case = oracle.OracleCase.from_input_dat(directory + "/input.dat", duplicate)
    rc, want_rms, iters = case.solve(cycles, indirect_rw)
    assert rc == 0
    for l in range(solver.num_levels):
        _assert_close(solver.get(l, "variables"), case.array(l, "variables").reshape(-1, 5), exact, f"level {l} variables after {cycles} cycles")
        _assert_close(solver.get(l, "residuals"), case.array(l, "residuals").reshape(-1, 5), exact, f"level {l} residuals")
        got = solver.loop_iters(l)
        want = {"flux": iters[l].flux, "update": 0, "compute_step": iters[l].compute_step,
                "time_step": iters[l].time_step, "restrict": iters[l].restrict_, "prolong": iters[l].prolong,
                "indirect_rw": iters[l].indirect_rw if indirect_rw else 0}
        assert got == want, f"level {l} LoopNumIters differ: {got} vs {want}"
    # RMS is summed in tree order on the GPU: printed with %.3e by the reference
    assert np.allclose(rms, want_rms, rtol=1e-12, atol=0)
    solver.close()


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("exact", [True, False])
def test_vcycles_three_levels(oracle, mesh3_dir, exact, fuse):
    import mgcfd
    _run_both(mgcfd, oracle, mesh3_dir, 4, exact, fuse=fuse)


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("variant", [1, 2, 3, 4, 16, 32])
def test_vcycles_flux_variants(oracle, mesh3_dir, variant, fuse):
    """Every flux variant (length factor streamed / recomputed, node gather / edge-once tiles / the
    two-phase design point) runs the same V-cycles bit-identically to the oracle."""
    import mgcfd
    _run_both(mgcfd, oracle, mesh3_dir, 4, True, fuse=fuse, variant=variant)
    _run_both(mgcfd, oracle, mesh3_dir, 2, False, fuse=fuse, variant=variant)


@pytest.mark.parametrize("variant", [2, 3])
def test_edge_once_on_ragged_tiles_and_fallback(oracle, variant):
    """Edge-once tiles on a random graph of degree 4: every tile's halo overflows the LDS tile, so
    end points come from the overflow table; each class on its own onto a non-zero flux array and all
    at once.  A degree-10 graph holds more edges per tile than the variant supports: the level must
    report that and the launch must fall back to the node gather — same bits either way."""
    import ctypes as C
    import mgcfd
    from mgcfd import meshgen
    lib = oracle.load()
    ff = oracle.farfield()
    for degree, expect in ((4, True), (10, False)):
        lvl = meshgen.make_random_graph_level(2500, degree=degree, seed=11 + degree)
        mg = meshgen.MultigridMesh(mesh_name="m6wing", levels=[lvl])
        L = mgcfd.generated_to_levels(mg)[0]
        s = mgcfd.Solver.from_arrays([L], mg.mesh_variant)
        assert s.has_edge_once(0) == expect
        s.set_option("flux_variant", variant)
        edges = np.ascontiguousarray(L["edges"]).copy()
        coords = np.ascontiguousarray(L["coords"], dtype=np.float64)
        lib.ora_adjust_ewt(oracle.ptr(coords), len(edges), oracle.ptr(edges))
        lib.ora_dampen_ewt(len(edges), oracle.ptr(edges), 5e-8)
        assert np.array_equal(s.get_edges(0, len(edges)), edges)
        ni, nb, nw = L["n_internal"], L["n_boundary"], L["n_wall"]
        q = perturbed_state(L["nel"], ff.var, seed=60 + degree)
        f0 = np.random.default_rng(degree).normal(size=(L["nel"], 5))
        s.set(0, "variables", q)
        s.set(0, "fluxes", f0)
        want = f0.copy()
        lib.ora_compute_flux_edge(0, ni, oracle.ptr(edges), oracle.ptr(q), oracle.ptr(want))
        s.compute_flux_edge(0)
        _assert_close(s.get(0, "fluxes"), want, True, f"degree {degree}: internal class onto a non-zero array")
        s.zero_fluxes(0)
        s.compute_fluxes(0)
        want = np.zeros_like(q)
        lib.ora_compute_flux_edge(0, ni, oracle.ptr(edges), oracle.ptr(q), oracle.ptr(want))
        lib.ora_compute_boundary_flux_edge(ni, nb, oracle.ptr(edges), oracle.ptr(q), oracle.ptr(want))
        lib.ora_compute_wall_flux_edge(ni + nb, nw, oracle.ptr(edges), oracle.ptr(q), oracle.ptr(want), C.byref(ff))
        _assert_close(s.get(0, "fluxes"), want, True, f"degree {degree}: all classes")
        # fused sweeps equal the node gather's (both from a zero flux array)
        s.zero_fluxes(0)
        s.set(0, "variables", q)
        s.smooth(0, 2)
        got = s.get(0, "variables")
        s.set_option("flux_variant", 0)
        s.zero_fluxes(0)
        s.set(0, "variables", q)
        s.smooth(0, 2)
        assert np.array_equal(got.view(np.int64), s.get(0, "variables").view(np.int64))
        s.close()


def test_vcycles_replayed_from_hipgraphs(oracle, mesh3_dir, fvcorr_dir):
    """MGCFD_OPT_GRAPH=1: whole cycles captured once per buffer-rotation / look-ahead state and replayed
    (7 cycles pass through every state of the 3-level case more than once)."""
    import mgcfd
    _run_both(mgcfd, oracle, mesh3_dir, 7, True, graph=1)
    _run_both(mgcfd, oracle, fvcorr_dir, 20, True, graph=1)


def test_vcycles_with_indirect_rw_and_duplication(oracle, mesh_dir):
    import mgcfd
    _run_both(mgcfd, oracle, mesh_dir, 3, True, indirect_rw=True, duplicate=2)


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("exact", [True, False])
def test_fvcorr_single_level(oracle, fvcorr_dir, exact, fuse):
    import mgcfd
    _run_both(mgcfd, oracle, fvcorr_dir, 50, exact, fuse=fuse)


def test_fvcorr_like_97k_nodes_developed_flow(oracle):
    """BASELINE configs[0] in its synthetic form: mesh_name = fvcorr, single level,
    46^3 box minus its centre node = 97,335 nodes, far-field outer faces + 6 solid-wall faces, permuted ids,
    undamped edge weights, Rodinia's local time step.  300 iterations (the flow develops: the RMS falls by
    an order of magnitude) against the oracle, bit for bit, then the reference's own -v tolerance rule."""
    import mgcfd
    from mgcfd import meshgen
    mg = meshgen.make_multigrid((46,), "fvcorr", seed=0, cavity_radius=0.001)
    assert mg.levels[0].nel == 46 ** 3 - 1
    levels = mgcfd.generated_to_levels(mg)
    assert levels[0]["n_boundary"] == 6
    iters = 300
    s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    rms = s.run_cycles(iters)
    got = s.get(0, "variables")
    assert s.loop_iters(0)["flux"] == 3 * iters * levels[0]["n_internal"]
    s.close()
    # oracle on the same arrays
    lib = oracle.load()
    L = levels[0]
    lv = (oracle.OraLevel * 1)()
    keep = [np.ascontiguousarray(L["volumes"], dtype=np.float64), np.ascontiguousarray(L["coords"], dtype=np.float64),
            np.ascontiguousarray(L["edges"]).copy()]
    state = [np.zeros((L["nel"], 5)) for _ in range(4)] + [np.zeros(L["nel"])]
    lv[0].nel, lv[0].n_edges = L["nel"], len(keep[2])
    lv[0].n_internal, lv[0].n_boundary, lv[0].n_wall = L["n_internal"], L["n_boundary"], L["n_wall"]
    lv[0].internal_start, lv[0].boundary_start, lv[0].wall_start = 0, L["n_internal"], L["n_internal"] + L["n_boundary"]
    lv[0].volumes, lv[0].coords, lv[0].edges = oracle.ptr(keep[0]), oracle.ptr(keep[1]), oracle.ptr(keep[2])
    lv[0].variables, lv[0].old_variables, lv[0].residuals, lv[0].fluxes = (oracle.ptr(a) for a in state[:4])
    lv[0].step_factors = oracle.ptr(state[4])
    want_rms = np.zeros(iters)
    assert lib.ora_solve(lv, 1, 0, iters, 0, oracle.ptr(want_rms), None) == 0
    _assert_close(got, state[0], True, "fvcorr-like 97K nodes, 300 iterations")
    assert np.allclose(rms, want_rms, rtol=1e-12, atol=0)
    assert rms[-1] < 0.2 * rms[0] and np.ptp(got[:, 0]) > 0.05          # a developed, converging flow
    assert lib.ora_identify_differences(oracle.ptr(np.ascontiguousarray(got)), oracle.ptr(state[0]), L["nel"], 0) == -1


def _oracle_solve_arrays(oracle, levels, mesh_variant, cycles):
    """ora_solve on in-memory level dicts (the generator's arrays): returns (per-level variables, rms, iters)."""
    lib = oracle.load()
    n = len(levels)
    lv = (oracle.OraLevel * n)()
    keep = []
    for l, L in enumerate(levels):
        vol = np.ascontiguousarray(L["volumes"], dtype=np.float64)
        coords = np.ascontiguousarray(L["coords"], dtype=np.float64)
        edges = np.ascontiguousarray(L["edges"]).copy()
        state = [np.zeros((L["nel"], 5)) for _ in range(4)] + [np.zeros(L["nel"])]
        keep.append((vol, coords, edges, state))
        lv[l].nel, lv[l].n_edges = L["nel"], len(edges)
        lv[l].n_internal, lv[l].n_boundary, lv[l].n_wall = L["n_internal"], L["n_boundary"], L["n_wall"]
        lv[l].internal_start, lv[l].boundary_start, lv[l].wall_start = 0, L["n_internal"], L["n_internal"] + L["n_boundary"]
        lv[l].volumes, lv[l].coords, lv[l].edges = oracle.ptr(vol), oracle.ptr(coords), oracle.ptr(edges)
        lv[l].variables, lv[l].old_variables, lv[l].residuals, lv[l].fluxes = (oracle.ptr(a) for a in state[:4])
        lv[l].step_factors = oracle.ptr(state[4])
        if L.get("mg_map") is not None and l + 1 < n:
            m = np.ascontiguousarray(L["mg_map"], dtype=np.int64)
            keep.append(m)
            lv[l].mg_map, lv[l].mgc = oracle.ptr(m), len(m)
    rms = np.zeros(cycles)
    iters = (oracle.OraIters * n)() if hasattr(oracle, "OraIters") else None
    rc = lib.ora_solve(lv, n, mesh_variant, cycles, 0, oracle.ptr(rms), iters)
    assert rc == 0
    return [k[3][0] for k in keep if isinstance(k, tuple)], rms


@pytest.mark.parametrize("sizes,name", [((2,), "fvcorr"), ((3,), "fvcorr"), ((5,), "m6wing"), ((6, 3), "m6wing"),
                                        ((7, 4, 2), "m6wing"), ((9, 5), "rotor37"), ((8, 4), "la_cascade"),
                                        ((17, 9, 5, 3), "m6wing")])
def test_tiny_and_odd_hierarchies(oracle, sizes, name):
    """Levels far smaller than a tile (8 nodes), node counts that are no multiple of 64, every mesh_name
    (damping 5e-8 / 1e-7 / 2e-7 / none), up to four levels: a few cycles, eager and graph-replayed, against
    the oracle bit for bit on every level."""
    import mgcfd
    from mgcfd import meshgen
    mg = meshgen.make_multigrid(sizes, name, seed=3, jitter=0.2, area_noise=0.05, volume_noise=0.05)
    levels = mgcfd.generated_to_levels(mg)
    cycles = 4
    want, want_rms = _oracle_solve_arrays(oracle, levels, mg.mesh_variant, cycles)
    for graph in (0, 1):
        s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
        s.set_option("graph", graph)
        rms = s.run_cycles(cycles)
        for l in range(len(levels)):
            _assert_close(s.get(l, "variables"), want[l], True, f"{sizes} {name} level {l} graph={graph}")
        assert np.allclose(rms, want_rms, rtol=1e-12, atol=0)
        s.close()


@pytest.mark.parametrize("name,sizes,cycles", [("m6wing", (6000, 1200, 300), 5), ("fvcorr", (6000,), 200),
                                               ("rotor37", (6000, 1000), 20)])
def test_unstructured_tetrahedral_median_dual_mesh(oracle, name, sizes, cycles):
    """Delaunay tetrahedralisations of random points with median-dual metrics (meshgen.make_tet_level): node
    degrees 6..40, 7.6 edges per node, hull nodes with solid-wall or far-field faces, ids without locality - the
    shape of the reference's real datasets rather than a lattice.  The undamped fvcorr case develops a flow
    (the wall faces push the state an O(1) distance from the far field) over 200 iterations.  Every level's state
    against the oracle bit for bit, for the default kernels, the k-recomputing rows and the edge-once tiles."""
    import mgcfd
    from mgcfd import meshgen
    mg = meshgen.make_tet_multigrid(sizes, name, seed=2)
    levels = mgcfd.generated_to_levels(mg)
    want, want_rms = _oracle_solve_arrays(oracle, levels, mg.mesh_variant, cycles)
    assert np.abs(want[0] - want[0][0]).max() > 1e-5            # not a uniform state
    for variant in (-1, 1, 2):
        s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
        s.set_option("flux_variant", variant)
        rms = s.run_cycles(cycles)
        for l in range(len(levels)):
            _assert_close(s.get(l, "variables"), want[l], True, f"tet {name} level {l} variant={variant}")
        assert np.allclose(rms, want_rms, rtol=1e-12, atol=0)
        s.close()


def _hand_level(nel, internal, faces, seed=0, scale=1e-3):
    """A level dict from explicit lists: internal = [(a, b)], faces = [(code, node)] with code -1 (solid wall) / -2 (far field)."""
    from mgcfd.meshgen import EDGE_DTYPE
    rng = np.random.default_rng(seed)
    faces = sorted(faces, key=lambda f: -f[0])                   # class order: internal | -1 | -2
    edges = np.zeros(len(internal) + len(faces), dtype=EDGE_DTYPE)
    for k, (a, b) in enumerate(internal):
        edges[k] = (a, b, *rng.normal(size=3) * scale)
    for k, (code, node) in enumerate(faces):
        edges[len(internal) + k] = (code, node, *rng.normal(size=3) * scale)
    return {"nel": nel, "volumes": rng.uniform(1e-6, 2e-6, nel), "coords": rng.random((nel, 3)), "edges": edges,
            "n_internal": len(internal), "n_boundary": sum(1 for c, _ in faces if c == -1),
            "n_wall": sum(1 for c, _ in faces if c == -2), "mg_map": None}


@pytest.mark.parametrize("name,nel,internal,faces", [
    ("one node, one far-field face", 1, [], [(-2, 0)]),
    ("one node, nothing attached", 1, [], []),
    ("three isolated nodes, wall faces only", 3, [], [(-1, 0), (-1, 2), (-1, 2)]),
    ("two nodes, one edge, no faces", 2, [(0, 1)], []),
    ("isolated nodes between connected ones", 6, [(0, 5), (2, 5)], [(-2, 3), (-1, 5), (-2, 5)]),
    ("a hub: one node on 300 edges (more rows than a tile has nodes)", 301, [(0, k) for k in range(1, 301)], [(-2, 0)]),
    ("the same pair joined by three edges, listed in both directions", 4, [(0, 1), (1, 0), (0, 1), (2, 3)], [(-2, 1)]),
    ("an edge from a node to itself", 3, [(0, 1), (1, 1), (1, 2), (2, 2)], [(-1, 2)]),
])
def test_degenerate_levels(oracle, name, nel, internal, faces):
    """Empty and degenerate inputs: levels without internal edges, without faces, with isolated nodes, with a node of
    degree 300 — a few iterations against the oracle, bit for bit, fused and kernel-by-kernel."""
    import mgcfd
    levels = [_hand_level(nel, internal, faces, seed=len(name), scale=1e-3 if nel < 100 else 1e-6)]   # (the hub: small weights, or the undamped state goes negative)
    want, want_rms = _oracle_solve_arrays(oracle, levels, 0, 3)          # fvcorr: undamped, local time step, no coords needed
    for fuse in (1, 0):
        s = mgcfd.Solver.from_arrays(levels, 0)
        s.set_option("fuse_update", fuse)
        rms = s.run_cycles(3)
        _assert_close(s.get(0, "variables"), want[0], True, f"{name} fuse={fuse}")
        assert np.allclose(rms, want_rms, rtol=1e-12, atol=1e-300)
        s.close()
    if any(a == b for a, b in internal):
        return          # (adjust_ewt divides by the distance between an edge's ends, validation.cpp:41-55: undefined for a self-edge)
    # the same with the global time step (m6wing: needs coordinates; damping 5e-8)
    want, want_rms = _oracle_solve_arrays(oracle, levels, 2, 2)
    s = mgcfd.Solver.from_arrays(levels, 2)
    s.run_cycles(2)
    _assert_close(s.get(0, "variables"), want[0], True, f"{name} m6wing")
    s.close()


@pytest.mark.parametrize("fuse", [1, 0])
def test_run_that_blows_up_reports_the_references_first_bad_cell(oracle, fuse):
    """check_for_invalid_variables (validation.cpp:107-138) runs after every time_step and exits at the first bad
    cell in original order.  An undamped hub with large weights goes negative after a few iterations: the library
    must return the same error class, name the same cell and the same cycle, and leave NaN in rms_out afterwards."""
    import mgcfd
    levels = [_hand_level(301, [(0, k) for k in range(1, 301)], [(-2, 0)], seed=5, scale=1.2e-4)]       # goes negative in cycle 7
    cycles = 12
    # the oracle stops inside the failing time_step and leaves the state as it was then
    lib = oracle.load()
    first_bad = None
    for c in range(1, cycles + 1):
        try:
            _oracle_solve_arrays(oracle, levels, 0, c)
        except AssertionError:
            first_bad = c - 1
            break
    assert first_bad is not None and first_bad >= 1, "the case must fail, and not in the very first cycle"
    n = len(levels)
    lv = (oracle.OraLevel * n)()
    L = levels[0]
    vol, coords, edges = (np.ascontiguousarray(L[k]).copy() for k in ("volumes", "coords", "edges"))
    state = [np.zeros((L["nel"], 5)) for _ in range(4)] + [np.zeros(L["nel"])]
    lv[0].nel, lv[0].n_edges, lv[0].n_internal, lv[0].n_boundary, lv[0].n_wall = L["nel"], len(edges), L["n_internal"], L["n_boundary"], L["n_wall"]
    lv[0].internal_start, lv[0].boundary_start, lv[0].wall_start = 0, L["n_internal"], L["n_internal"] + L["n_boundary"]
    lv[0].volumes, lv[0].coords, lv[0].edges = oracle.ptr(vol), oracle.ptr(coords), oracle.ptr(edges)
    lv[0].variables, lv[0].old_variables, lv[0].residuals, lv[0].fluxes = (oracle.ptr(a) for a in state[:4])
    lv[0].step_factors = oracle.ptr(state[4])
    rms = np.zeros(cycles)
    want_code = lib.ora_solve(lv, 1, 0, cycles, 0, oracle.ptr(rms), None)
    assert want_code in (1, 2, 3)
    bad = C.c_int64(-1)
    assert lib.ora_check_for_invalid_variables(oracle.ptr(state[0]), L["nel"], C.byref(bad)) == want_code
    s = mgcfd.Solver.from_arrays(levels, 0)
    s.set_option("fuse_update", fuse)
    out = np.zeros(cycles)
    rc = s.lib.mgcfd_run_cycles(s.handle, cycles, out.ctypes.data_as(C.c_void_p))
    assert rc == {1: 4, 2: 5, 3: 6}[want_code]                  # MGCFD_ERR_NAN / NEG_DENSITY / NEG_ENERGY
    assert s.invalid_state_location() == (bad.value, first_bad)
    assert np.isfinite(out[:first_bad]).all() and np.isnan(out[first_bad:]).all()
    assert f"at cell {bad.value} in cycle {first_bad + 1}" in s.lib.mgcfd_last_error().decode()
    s.close()


@pytest.mark.parametrize("seed", [0, 3, 7, 12, 19, 23, 40, 57, 88, 101, 127, 149])
def test_randomised_hierarchies(oracle, seed):
    """tools/fuzz_parity.py's generator (random kind — lattice, tetrahedra, hub, random graph —, sizes, mesh name,
    options; 600 seeds ran clean when it was written): a dozen of them here, every level bit for bit.  Seed 40 is a
    hierarchy whose COARSE level is the larger one."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity
    import mgcfd
    rng = np.random.default_rng(seed)
    kind, name, mg, cycles = fuzz_parity.make_case(rng)
    levels = mgcfd.generated_to_levels(mg)
    want, want_rms = _oracle_solve_arrays(oracle, levels, mg.mesh_variant, cycles)
    s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    s.set_option("fuse_update", int(rng.integers(0, 2)))
    s.set_option("flux_variant", int(rng.choice([-1, 0, 1, 2, 3, 16, 32])))
    s.set_option("graph", int(rng.integers(0, 2)))
    rms = s.run_cycles(cycles)
    for l in range(len(levels)):
        _assert_close(s.get(l, "variables"), want[l], True, f"seed {seed}: {kind} {name} level {l}")
    assert np.allclose(rms, want_rms, rtol=1e-12, atol=1e-300)
    s.close()


@pytest.mark.parametrize("seed", [2, 7, 12, 22, 157, 159, 222, 392, 640, 901])
def test_randomised_operation_sequences(seed):
    """tools/fuzz_ops.py: kernel-granular calls, sweeps (plain, split around the collective hooks), cycles, transfers,
    array writes and option changes in random order, mirrored call by call on the oracle; every array of every level is
    compared bit for bit after every call.  1,600 seeds ran clean when this was written (an earlier version of the
    generator found the bug test_graph_replay_never_covers_unfused_sweeps pins)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_ops
    log = fuzz_ops.run_seed(seed, 60)
    assert len(log) >= 2


@pytest.mark.gpu
def test_half_rows_never_exceed_what_a_lane_keeps():
    """tools/fuzz_partitioned.py seed 2070: a part of a small tetrahedral level — 87 owned nodes of degree up to 17 in one
    tile, no long rows — whose first slice owns eleven evaluations per lane.  The half-row plan used to give that slice
    eleven half rows although k_flux_half keeps five per lane (the rest were never evaluated: garbage fluxes); now the
    surplus goes to the other slices' lanes or the level runs the node gather.  Half rows (variant 32) must write the node
    gather's bits on every part."""
    import os
    import sys
    import mgcfd
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity
    from mgcfd.partition import partition_level, rcb_partition
    rng = np.random.default_rng(5000 + 2070)
    while True:
        kind, name, mg, _ = fuzz_parity.make_case(rng)
        if name != "fvcorr" and 20 <= mg.levels[0].nel <= 5000:
            break
    assert kind == "tet" and mg.levels[0].nel == 260
    L = mgcfd.generated_to_levels(mg)[0]
    parts = partition_level(L, rcb_partition(np.asarray(L["coords"]), 3))
    ff = None
    with_half = 0
    for P in parts:
        out = {}
        for v in (0, 32):
            s = mgcfd.Solver.from_arrays([P.level], mg.mesh_variant, n_owned=[P.n_owned])
            ff = s.far_field()[:5]
            s.set_option("flux_variant", v)
            s.set(0, "variables", perturbed_state(L["nel"], ff, seed=2070)[P.global_ids])
            with_half += int(v == 32 and s.has_half_rows(0))
            s.compute_fluxes(0)
            out[v] = s.get(0, "fluxes")[:P.n_owned]
            s.close()
        assert np.array_equal(out[0].view(np.int64), out[32].view(np.int64)), P.rank
    # (with the cap in place none of these three parts has room for its evaluations in five half rows per lane, so all
    #  of them decline and run the node gather: with_half == 0; before the fix part 2 claimed half rows and wrote garbage)
    assert with_half <= len(parts)


def test_graph_replay_never_covers_unfused_sweeps(mesh3_dir):
    """MGCFD_OPT_GRAPH=1 with the two-phase flux variant (whose sweeps run unfused): a replayed sweep used to leave the
    host-side "fluxes are stale" flag unset, so mgcfd_get_array(fluxes) returned the last stage's fluxes instead of
    the zeros time_step leaves (cfd_loops.cpp:266-268).  Sweeps and cycles, several replays each."""
    import mgcfd
    want = mgcfd.Solver.from_mesh(mgcfd.Mesh("input.dat", mesh3_dir))
    s = mgcfd.Solver.from_mesh(mgcfd.Mesh("input.dat", mesh3_dir))
    s.set_option("graph", 1)
    s.set_option("flux_variant", 4)
    q = perturbed_state(s.nel(0), s.far_field()[:5], seed=3)
    for x in (s, want):
        x.set(0, "variables", q)
    for k in range(4):
        s.smooth(0, 1); want.smooth(0, 1)
        assert not s.get(0, "fluxes").any(), f"sweep {k}"
        assert np.array_equal(s.get(0, "variables").view(np.int64), want.get(0, "variables").view(np.int64))
    for k in range(3):
        s.run_cycles(1); want.run_cycles(1)
        for l in range(s.num_levels):
            assert not s.get(l, "fluxes").any(), f"cycle {k} level {l}"
            assert np.array_equal(s.get(l, "variables").view(np.int64), want.get(l, "variables").view(np.int64))
    s.close(); want.close()


def test_tiling_report_and_coordinate_box_fallback():
    """mgcfd_level_tiling: a lattice level keeps the greedy clusters (no halo node left outside the LDS tile); on a
    tetrahedral level those overflow, coordinate boxes are chosen instead, and few row entries are left to gather
    from HBM."""
    import mgcfd
    from mgcfd import meshgen
    mg = meshgen.make_multigrid((20,), "m6wing", seed=1, jitter=0.2)
    s = mgcfd.Solver.from_arrays(mgcfd.generated_to_levels(mg), mg.mesh_variant)
    t = s.tiling(0)
    assert t["tiles"] == (8000 + 255) // 256 and t["overflow_refs"] == 0 and t["coordinate_boxes"] == 0
    assert t["halo_max"] <= t["halo_capacity"] == 303
    assert t["row_entries"] == 2 * s.num_internal_edges(0) and t["list_entries"] == 0
    s.close()
    mg = meshgen.make_tet_multigrid((30000,), "m6wing", seed=1)
    s = mgcfd.Solver.from_arrays(mgcfd.generated_to_levels(mg), mg.mesh_variant)
    t = s.tiling(0)
    assert t["coordinate_boxes"] == 1 and t["halo_max"] > t["halo_capacity"]
    assert t["row_entries"] == 2 * s.num_internal_edges(0)
    assert t["overflow_refs"] < 0.02 * t["row_entries"]
    # long rows: the highest-degree nodes' last entries (and every entry after an unstaged neighbour) go to the workgroups
    assert t["overflow_refs"] <= t["list_entries"] < 0.2 * t["row_entries"]
    s.close()


@pytest.mark.parametrize("fuse", [1, 0])
def test_old_variables_hold_the_sweep_start_state(mesh3_dir, fvcorr_dir, fuse):
    """The fused sweeps never copy variables to old_variables (the three state buffers change roles instead) and may
    skip compute_step_factor's kernel because an earlier launch looked ahead: after every sweep old_variables must
    still be the state the sweep started from, residuals = variables - old_variables, and writing variables from
    outside must invalidate whatever was computed ahead."""
    import mgcfd
    for directory in (mesh3_dir, fvcorr_dir):
        s = mgcfd.Solver.from_mesh(mgcfd.Mesh("input.dat", directory))
        s.set_option("fuse_update", fuse)
        q = perturbed_state(s.nel(0), s.far_field()[:5], seed=21)
        s.set(0, "variables", q)
        prev = q
        for _ in range(4):                                  # more than one period of the buffer rotation
            s.smooth(0, 1)
            cur, old, res = s.get(0, "variables"), s.get(0, "old_variables"), s.get(0, "residuals")
            assert np.array_equal(old.view(np.int64), prev.view(np.int64))
            assert np.array_equal(res.view(np.int64), (cur - old).view(np.int64))
            prev = cur
        # a state written from outside: the next sweep must recompute its step factors from it
        q2 = perturbed_state(s.nel(0), s.far_field()[:5], seed=22)
        s.set(0, "variables", q2)
        s.smooth(0, 1)
        fresh = mgcfd.Solver.from_mesh(mgcfd.Mesh("input.dat", directory))
        fresh.set_option("fuse_update", fuse)
        fresh.set(0, "variables", q2)
        fresh.smooth(0, 1)
        assert np.array_equal(s.get(0, "variables").view(np.int64), fresh.get(0, "variables").view(np.int64))
        assert np.array_equal(s.get(0, "step_factors").view(np.int64), fresh.get(0, "step_factors").view(np.int64))
        s.close(); fresh.close()


@pytest.mark.gpu
@pytest.mark.parametrize("then", ["set_variables", "time_step", "copy_old", "unfused_sweep", "split_sweep", "cycles", "graph_sweep",
                                  "written_from_outside", "bench_flux", "cycle_graph_replay", "nothing"])
def test_unwritten_residual_is_written_before_its_operands_change(fvcorr_dir, then, monkeypatch):
    """On a single-level run the last stage of a fused sweep does not write residuals[] (the next sweep would overwrite it
    unread); the library writes it on demand from `variables` and the sweep's start state.  Whatever changes either
    operand first must therefore write it first: the same call sequence on a solver that writes every residual
    (MGCFD_LAZY_RESIDUAL=0) must leave every array bit-identical, the residual included, read only at the very end."""
    import mgcfd
    from mgcfd import meshgen
    cases = [lambda: mgcfd.Solver.from_mesh(mgcfd.Mesh("input.dat", fvcorr_dir)),                 # local time step
             lambda: mgcfd.Solver.from_generated(meshgen.make_multigrid((11,), "m6wing", seed=4, cavity_radius=0.2))]   # global time step
    for make in cases:
        solvers = []
        for lazy in ("1", "0"):
            monkeypatch.setenv("MGCFD_LAZY_RESIDUAL", lazy)
            solvers.append(make())
        monkeypatch.delenv("MGCFD_LAZY_RESIDUAL")
        q = perturbed_state(solvers[0].nel(0), solvers[0].far_field()[:5], seed=31)
        q2 = perturbed_state(solvers[0].nel(0), solvers[0].far_field()[:5], seed=32)
        out = []
        for s in solvers:
            s.set(0, "variables", q)
            s.smooth(0, 2)                                  # (nothing read in between: the second sweep's residual is unwritten)
            if then == "set_variables":
                s.set(0, "variables", q2)
            elif then == "time_step":
                s.compute_fluxes(0); s.time_step(0, 1)
            elif then == "copy_old":
                s.copy_old_variables(0)
            elif then == "unfused_sweep":
                s.copy_old_variables(0); s.compute_step_factor(0)
                for j in range(3):
                    s.compute_fluxes(0); s.time_step(0, j)
            elif then == "split_sweep":
                s.sweep_begin(0); s.sweep_stage(0, 0, partials=False); s.sweep_stage(0, 1, partials=False)     # (stops before the last stage)
            elif then == "cycles":
                s.run_cycles(2)
            elif then == "graph_sweep":
                s.set_option("graph", 1); s.smooth(0, 4)
            elif then == "written_from_outside":
                s.array_devptr(0, "variables"); s.array_written(0, "variables")
            elif then == "bench_flux":
                s.bench_flux(0, 2); s.zero_fluxes(0)
            elif then == "cycle_graph_replay":
                # cycles captured into graphs (their last stage leaves the residual unwritten), a sweep that WRITES its residual,
                # then the same graphs replayed: the replay must mark the residual unwritten again (tools/fuzz_ops.py seed 1082)
                s.set_option("graph", 1); s.run_cycles(3)
                s.sweep_begin(0)
                for j in range(3):
                    s.sweep_stage(0, j, partials=False)
                s.run_cycles(3)
            names = ["residuals", "variables", "old_variables", "step_factors"]
            if then not in ("split_sweep",):
                names.append("fluxes")
            out.append({n: s.get(0, n) for n in names})
            s.close()
        for n in out[0]:
            assert np.array_equal(out[0][n].view(np.int64), out[1][n].view(np.int64)), (then, n)
        assert np.abs(out[0]["residuals"]).max() > 0.0


def test_invalid_state_is_reported(setup, oracle):
    mgcfd, mesh, solver, case, lib = setup
    ff = oracle.farfield()
    q = perturbed_state(case.levels[0].nel, ff.var, seed=1)
    q[17, 0] = -1.0
    q[40, 4] = np.nan
    solver.set(0, "variables", q)
    rc, bad = solver.check_for_invalid_variables(0)
    # the reference reports the first offending cell in original order: cell 17, negative density
    assert (rc, bad) == (5, 17)
    solver.set(0, "variables", perturbed_state(case.levels[0].nel, ff.var, seed=1))
    assert solver.check_for_invalid_variables(0)[0] == 0


# ------------------------------------------------------------------------------------------
# Golden fixtures (outputs of the REAL reference, tests/golden/) through the product
# ------------------------------------------------------------------------------------------
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = ["m6_2lvl", "m6_3lvl", "m6_2lvl_dup2", "fvcorr_1lvl", "tet_2lvl", "mixed_2lvl"]


def _case(name):
    d = os.path.join(GOLDEN, name)
    meta = dict(l.strip().split(" = ") for l in open(os.path.join(d, "case.txt")))
    return d, int(meta["cycles"]), int(meta["duplicate"])


def _csv_row(path):
    rows = [l.rstrip(",\n").split(",") for l in open(path) if l.strip()]
    return dict(zip(rows[0], rows[1]))


@pytest.mark.parametrize("mode", ["timers", "loop-timers", "no-timers"])
@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_driver_reproduces_reference_binary(case, mode, tmp_path):
    """euler3d_gpu_double with the reference's command line: the variables dump must be
    byte-identical to the reference's, LoopNumIters.csv must carry the same counts.  "timers" is the default run: fused
    stages, the per-loop times of Times.csv attributed from every 32nd sweep (MGCFD_OPT_TIMING = 4); "loop-timers" brackets
    every loop as the reference's -DTIME build does; "no-timers" measures nothing but the total."""
    d, cycles, dup = _case(case)
    exe = os.path.join(ROOT, "mg-cfd-app-plain_amd", "csrc", "euler3d_gpu_double")
    cmd = [exe, "-i", "input.dat", "-d", os.path.join(d, "input"), "-o", str(tmp_path) + "/", "-g", str(cycles),
           "-m", str(dup), "--output-variables"]
    if mode != "timers":
        cmd.append("--" + mode)
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    dump = tmp_path / f"variables.size={dup}x.cycles={cycles}.level=0"
    assert dump.read_bytes() == open(os.path.join(d, "variables.level0.txt"), "rb").read()
    want_lines = [l.strip() for l in open(os.path.join(d, "stdout.txt")) if "RMS" in l]
    got_lines = [l.strip() for l in r.stdout.splitlines() if "RMS" in l]
    assert got_lines == want_lines
    assert "Total runtime = " in r.stdout and "Loop stats written to:" in r.stdout
    want, got = _csv_row(os.path.join(d, "LoopNumIters.csv")), _csv_row(tmp_path / "LoopNumIters.csv")
    assert list(got.keys()) == list(want.keys())                      # same schema, same column order
    for k in want:
        if k[:-1] in ("flux", "update", "compute_step", "time_step", "restrict", "prolong") or k in ("Size", "Mesh", "MG cycles"):
            assert got[k] == want[k], k
        if k.startswith("indirect_rw"):
            assert got[k] == (want[k] if mode != "no-timers" else "0")    # the probe is skipped on the fast path
    t = _csv_row(tmp_path / "Times.csv")
    assert list(t.keys())[:18] == list(want.keys())[:18] and "Total" in t
    if mode != "no-timers":
        # every loop the reference times has a time, on every level; together they do not exceed the run
        levels = sum(1 for k in want if k.startswith("flux"))
        loops = ["flux", "compute_step", "time_step", "indirect_rw"] + (["restrict"] if levels > 1 else [])
        for l in range(levels):
            for name in loops + (["prolong"] if l + 1 < levels else []):
                if name == "restrict" and l == 0:
                    continue                                               # (booked on the coarse level, as the reference does)
                assert float(t[f"{name}{l}"]) > 0, f"{name}{l}"
        # (indirect_rw aside: the attributed run extrapolates the probe's column at its measured rate)
        assert sum(float(v) for k, v in t.items() if k[:-1] in ("flux", "compute_step", "time_step", "restrict", "prolong")) <= float(t["Total"]) * 1.02
        assert ("attributed" in t["Flux options"]) == (mode == "timers")


@pytest.mark.parametrize("case", ["fvcorr_hub_nan", "fvcorr_hub_negative_energy"])
@pytest.mark.parametrize("extra", [[], ["--no-timers"]])
def test_driver_aborts_like_the_reference_binary(case, extra, tmp_path):
    """Runs the reference aborted in check_for_invalid_variables (golden stdout of the real binary): same cycle
    lines (the failing cycle's without an RMS), same ERROR line, same first bad cell, non-zero exit code."""
    d, cycles, _ = _case(case)
    want = [l.rstrip("\n") for l in open(os.path.join(d, "stdout.txt"))]
    exe = os.path.join(ROOT, "mg-cfd-app-plain_amd", "csrc", "euler3d_gpu_double")
    r = subprocess.run([exe, "-i", "input.dat", "-d", os.path.join(d, "input"), "-o", str(tmp_path) + "/", "-g", str(cycles)] + extra,
                       capture_output=True, text=True)
    assert r.returncode == 1
    got = [l for l in r.stdout.splitlines() if l.startswith("Cycle") or l.startswith("ERROR") or l.startswith("Cell")]
    k = next(i for i, l in enumerate(want) if l.startswith("ERROR"))
    assert got[:k + 1] == want[:k + 1]                                 # cycle lines with their RMS values, ERROR line
    assert len(got) == k + 2 and want[k + 1].startswith(got[k + 1] + ":")   # "Cell 0" (the reference adds the cell's values)
    assert "Total runtime" not in r.stdout


def test_driver_randomised_inputs_and_flags():
    """tools/fuzz_driver.py: random hierarchies written in the reference's file formats, the drop-in binary with random
    -g / -m / --no-timers / --no-indirect-rw / --legacy-ordering, against the oracle reading the same files (dump bit
    for bit, RMS lines, LoopNumIters), several ranks (--gpus N, all on this GPU) and the reference's own main() on the
    library included.  Some of its seeds; 10105 is a hierarchy whose last prolongation spoils a value: the reference
    (which checks after a time_step only) ends normally, and so must the drop-in with one level per rank."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_driver
    assert fuzz_driver.run_seeds(0, 3) == 0
    assert fuzz_driver.run_seeds(40, 3) == 0
    assert fuzz_driver.run_seeds(10105, 1) == 0


def test_driver_legacy_ordering_reproduces_reference_built_with_that_flag(tmp_path):
    """--legacy-ordering == the reference compiled with -DLEGACY_ORDERING (src/Base/io.cpp:183-193): byte-identical
    dump; and the Python binding's Mesh(legacy_ordering=True) hands back the sorted edges."""
    d, cycles, dup = _case("fvcorr_1lvl_legacy_ordering")
    exe = os.path.join(ROOT, "mg-cfd-app-plain_amd", "csrc", "euler3d_gpu_double")
    for extra in ([], ["--no-timers"]):
        out = tmp_path / ("a" if not extra else "b")
        out.mkdir()
        r = subprocess.run([exe, "-i", "input.dat", "-d", os.path.join(d, "input"), "-o", str(out) + "/", "-g", str(cycles),
                            "--output-variables", "--legacy-ordering"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        dump = out / f"variables.size={dup}x.cycles={cycles}.level=0"
        assert dump.read_bytes() == open(os.path.join(d, "variables.level0.txt"), "rb").read()
    import mgcfd
    m = mgcfd.Mesh("input.dat", os.path.join(d, "input"), legacy_ordering=True)
    lv = m.level(0)
    ni = lv["n_internal"]
    a, b = lv["edges"]["a"][:ni], lv["edges"]["b"][:ni]
    assert np.all((np.diff(a) > 0) | ((np.diff(a) == 0) & (np.diff(b) >= 0)))


def test_driver_validate_result_and_config_file(tmp_path):
    """-v against a solution.* file (src/euler3d_cpu_double.cpp:704-744, tolerance rule of
    validation.cpp:140-199) and the -c key=value config file (src/Base/config.cpp:159-217)."""
    import shutil
    d, cycles, dup = _case("m6_2lvl")
    work = tmp_path / "in"
    shutil.copytree(os.path.join(d, "input"), work)
    sol = work / f"solution.variables.size={dup}x.cycles={cycles}.level=0"
    shutil.copy(os.path.join(d, "variables.level0.txt"), sol)
    exe = os.path.join(ROOT, "mg-cfd-app-plain_amd", "csrc", "euler3d_gpu_double")
    out = tmp_path / "out"
    out.mkdir()
    base = [exe, "-i", "input.dat", "-d", str(work), "-o", str(out) + "/", "-g", str(cycles), "-v"]
    r = subprocess.run(base, capture_output=True, text=True)
    assert r.returncode == 0 and "NaN check passed" in r.stdout and "PASS: variables[] validated successfully" in r.stdout
    # FMA-contracted kernels still validate (differences ~1e-22 against a floor of 3e-19)
    r = subprocess.run(base + ["--fast", "--no-timers"], capture_output=True, text=True)
    assert r.returncode == 0 and "PASS: variables[] validated successfully" in r.stdout
    # a wrong solution file must fail the run, as the reference's exit(EXIT_FAILURE) does
    vals = np.loadtxt(sol)
    vals[5, 2] += 1e-9
    np.savetxt(sol, vals, fmt="%.17e")
    r = subprocess.run(base, capture_output=True, text=True)
    assert r.returncode != 0 and "ERROR: Unacceptable error detected at (i=5, v=2)" in r.stdout
    # missing solution file: message, no PASS line, exit code 0 (euler3d_cpu_double.cpp:718-725)
    os.remove(sol)
    r = subprocess.run(base, capture_output=True, text=True)
    assert r.returncode == 0 and "aborting validation" in r.stdout and "PASS" not in r.stdout
    # -c: the same run described by a config file placed next to the inputs
    (work / "run.conf").write_text(f"# config\ninput_file = input.dat\ninput_file_directory = ./\ncycles = {cycles}\n"
                                   f"output_file_prefix = {out}/cfg\noutput_variables = Y\nmesh_duplicate_count = 1\n")
    r = subprocess.run([exe, "-c", str(work / "run.conf")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    dump = out / f"cfg.variables.size=1x.cycles={cycles}.level=0"           # prefix without '/' => '<prefix>.<name>'
    assert dump.read_bytes() == open(os.path.join(d, "variables.level0.txt"), "rb").read()
    assert (out / "cfg.Times.csv").exists() and (out / "cfg.LoopNumIters.csv").exists()


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_kernels_reproduce_reference_vectors(case):
    import mgcfd
    d, cycles, dup = _case(case)
    g = np.load(os.path.join(d, "kernels.npz"))
    mesh = mgcfd.Mesh("input.dat", os.path.join(d, "input"))
    s = mgcfd.Solver.from_mesh(mesh)
    assert np.array_equal(s.far_field(), g["far_field"])
    for l in range(s.num_levels):
        assert np.array_equal(s.get_edges(l, len(g[f"L{l}_edges"])), g[f"L{l}_edges"])
        s.set(l, "variables", g[f"L{l}_q"])
        s.set(l, "fluxes", g[f"L{l}_flux_in"])
        s.compute_flux_edge(l)
        _assert_close(s.get(l, "fluxes"), g[f"L{l}_flux_internal"], True, "internal flux")
        s.compute_boundary_flux_edge(l)
        _assert_close(s.get(l, "fluxes"), g[f"L{l}_flux_boundary"], True, "boundary flux")
        s.compute_wall_flux_edge(l)
        _assert_close(s.get(l, "fluxes"), g[f"L{l}_flux_wall"], True, "wall flux")
        s.zero_fluxes(l)
        s.indirect_rw(l)
        _assert_close(s.get(l, "fluxes"), g[f"L{l}_indirect_rw"], True, "indirect_rw")
        s.compute_step_factor(l)
        _assert_close(s.get(l, "step_factors"), g[f"L{l}_step_factors"], True, "step factors")
        for j in range(3):
            s.set(l, "fluxes", g[f"L{l}_ts_flux"])
            s.set(l, "old_variables", g[f"L{l}_ts_old"])
            s.time_step(l, j)
            _assert_close(s.get(l, "variables"), g[f"L{l}_ts_j{j}"], True, f"time_step {j}")
        s.set(l, "variables", g[f"L{l}_q"])
        s.residual(l)
        _assert_close(s.get(l, "residuals"), g[f"L{l}_residual"], True, "residual")
        assert abs(s.calc_rms(l) - g[f"L{l}_rms"][0]) <= 1e-13 * g[f"L{l}_rms"][0]
    for l in range(s.num_levels - 1):
        s.set(l, "variables", g[f"T{l}_qf"])
        s.set(l + 1, "variables", g[f"T{l}_qc"])
        s.restrict(l)
        _assert_close(s.get(l + 1, "variables"), g[f"T{l}_restrict"], True, "restrict")
        s.set(l, "variables", g[f"T{l}_qf"])
        s.set(l, "residuals", g[f"T{l}_rf"])
        s.set(l + 1, "residuals", g[f"T{l}_rc"])
        s.prolong(l)
        _assert_close(s.get(l, "variables"), g[f"T{l}_prolong"], True, "prolong")
    s.close()


def test_sharded_sweep_hooks_equal_smooth(mesh3_dir):
    """The multi-GPU sweep (split around the all-reduce) run on one rank must equal mgcfd_smooth."""
    import torch
    import mgcfd
    from mgcfd.distributed import HipSolverAdapter, ShardedSweep
    mesh = mgcfd.Mesh("input.dat", mesh3_dir)
    results = []
    for mode in ("smooth", "sharded-fused", "sharded-fused-overlap", "sharded-fused-scalar", "sharded-fused-scalar-overlap",
                 "sharded-unfused"):
        s = mgcfd.Solver.from_mesh(mesh)
        q = perturbed_state(s.nel(0), s.far_field()[:5], seed=77)
        s.set(0, "variables", q)
        if mode == "smooth":
            s.smooth(0, 3)
            rms = s.calc_rms(0)
        else:
            st = torch.cuda.Stream()
            torch.cuda.set_stream(st)
            s.set_stream(st.cuda_stream)
            sw = ShardedSweep(HipSolverAdapter(s, torch.device("cuda", 0)), None, fused=mode.startswith("sharded-fused"))
            sw.overlap_even_alone = mode.endswith("overlap")      # the path a multi-rank run takes
            sw.reduce_partials = "scalar" not in mode             # default: the partial minima are what is all-reduced
            for _ in range(3):
                sw.sweep(0)
            rms = sw.rms(0, s.nel(0))
        torch.cuda.set_stream(torch.cuda.default_stream())
        results.append((s.get(0, "variables"), s.get(0, "residuals"), s.get(0, "step_factors"), rms))
        s.close()
    for other in results[1:]:
        for a, b in zip(results[0][:3], other[:3]):
            assert np.array_equal(a.view(np.int64), b.view(np.int64))
        assert abs(other[3] - results[0][3]) <= 1e-13 * results[0][3]


@pytest.mark.parametrize("exact", [True, False])
def test_ragged_tiles_overflow_the_lds_halo(oracle, exact):
    """A random (non-geometric) graph: every tile's halo is far larger than the LDS tile holds, so
    most neighbour reads take the overflow path (straight from HBM).  Also a level with isolated
    nodes, nodes with many faces, and a fine->coarse map that leaves coarse nodes without children."""
    import ctypes as C
    import mgcfd
    from mgcfd import meshgen
    fine = meshgen.make_random_graph_level(3000, degree=8, seed=1)
    coarse = meshgen.make_random_graph_level(500, degree=6, seed=2)
    fine.mg_map = (np.random.default_rng(3).integers(0, 400, fine.nel)).astype(np.int64)   # coarse 400..499 unmapped
    mg = meshgen.MultigridMesh(mesh_name="m6wing", levels=[fine, coarse])
    levels = mgcfd.generated_to_levels(mg)
    s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    s.set_option("exact", int(exact))
    lib = oracle.load()
    ff = oracle.farfield()
    for l, L in enumerate(levels):
        edges = np.ascontiguousarray(L["edges"]).copy()
        coords = np.ascontiguousarray(L["coords"], dtype=np.float64)
        lib.ora_adjust_ewt(oracle.ptr(coords), len(edges), oracle.ptr(edges))
        lib.ora_dampen_ewt(len(edges), oracle.ptr(edges), 5e-8)
        assert np.array_equal(s.get_edges(l, len(edges)), edges)
        ni, nb, nw = L["n_internal"], L["n_boundary"], L["n_wall"]
        q = perturbed_state(L["nel"], ff.var, seed=40 + l)
        want = np.zeros_like(q)
        lib.ora_compute_flux_edge(0, ni, oracle.ptr(edges), oracle.ptr(q), oracle.ptr(want))
        lib.ora_compute_boundary_flux_edge(ni, nb, oracle.ptr(edges), oracle.ptr(q), oracle.ptr(want))
        lib.ora_compute_wall_flux_edge(ni + nb, nw, oracle.ptr(edges), oracle.ptr(q), oracle.ptr(want), C.byref(ff))
        s.set(l, "variables", q)
        s.zero_fluxes(l)
        s.compute_fluxes(l)
        _assert_close(s.get(l, "fluxes"), want, exact, f"random-graph level {l} fluxes")
    # restriction with unmapped coarse nodes keeps their value (mg_loops.cpp:63-78,174-189)
    qf = perturbed_state(fine.nel, ff.var, seed=50)
    qc = perturbed_state(coarse.nel, ff.var, seed=51)
    want = qc.copy()
    scratch = np.zeros(coarse.nel, dtype=np.int64)
    lib.ora_mg_restrict(oracle.ptr(qf), oracle.ptr(want), coarse.nel, oracle.ptr(fine.mg_map), oracle.ptr(scratch), fine.nel)
    s.set(0, "variables", qf)
    s.set(1, "variables", qc)
    s.restrict(0)
    got = s.get(1, "variables")
    _assert_close(got, want, exact, "restrict with unmapped coarse nodes")
    assert np.array_equal(got[400:], qc[400:])
    # a fused sweep on the ragged level equals the kernel-granular one
    for fuse in (1, 0):
        s.set_option("fuse_update", fuse)
        s.set(0, "variables", qf)
        s.zero_fluxes(0)
        s.smooth(0, 2)
        if fuse:
            ref_v = s.get(0, "variables")
        else:
            assert np.array_equal(s.get(0, "variables").view(np.int64), ref_v.view(np.int64))
    s.close()


@pytest.mark.parametrize("variant,n_parts,partitioner,fused,mesh", [
    (0, 3, "slab", False, "lattice"), (2, 3, "slab", False, "lattice"), (0, 4, "rcb", False, "lattice"),
    (0, 3, "slab", True, "lattice"), (2, 4, "rcb", True, "lattice"),
    (-1, 3, "rcb", True, "tet"), (-1, 4, "slab", False, "tet"),       # tetrahedra: long rows and unstaged neighbours in every part
    (-1, 3, "slab", True, "hub"), (-1, 2, "rcb", False, "hub"),       # a 700-spoke hub: owned by one part, a ghost in the others
    (-1, 3, "rcb", True, "graph"), (-1, 3, "slab", False, "graph")])   # random graph: nearly every node of a part has ghosts
def test_partitioned_level_with_halo_exchange_equals_whole_mesh(variant, n_parts, partitioner, fused, mesh):
    """BASELINE config 5 in miniature: one level split into 3 parts with ghost nodes, every RK stage
    followed by a halo exchange (packed / unpacked on the GPU), global-min time step over all parts.
    The three parts run as three solvers on this one GPU, threads standing in for ranks and an
    in-process copy for the RCCL send/recv; owned nodes must equal the unpartitioned run bit for bit."""
    import threading
    import torch
    import mgcfd
    from mgcfd import meshgen
    from mgcfd.distributed import HipSolverAdapter, PartitionedSweep
    from mgcfd.partition import partition_level, rcb_partition, slab_partition
    dev = torch.device("cuda", 0)
    if mesh == "tet":
        mg = meshgen.make_tet_multigrid((5000,), "m6wing", seed=6)
    elif mesh == "hub":
        mg = meshgen.MultigridMesh(mesh_name="m6wing", levels=[meshgen.make_hub_level(700, scale=1e-4, seed=6)])
    elif mesh == "graph":
        mg = meshgen.MultigridMesh(mesh_name="m6wing", levels=[meshgen.make_random_graph_level(2500, degree=8, seed=6)])
    else:
        mg = meshgen.make_multigrid((14,), "m6wing", seed=4, cavity_radius=0.15, jitter=0.2, area_noise=0.05, volume_noise=0.05)
    _partitioned_level_check(mg, variant, n_parts, partitioner, fused)


def _partitioned_level_check(mg, variant, n_parts, partitioner, fused, sweeps=3, seed=9):
    """Level 0 of `mg` split into n_parts solvers on this GPU against the unpartitioned run (see the test above)."""
    import threading
    import torch
    import mgcfd
    from mgcfd.distributed import HipSolverAdapter, PartitionedSweep
    from mgcfd.partition import partition_level, rcb_partition, slab_partition
    dev = torch.device("cuda", 0)
    L = mgcfd.generated_to_levels(mg)[0]
    split = slab_partition if partitioner == "slab" else rcb_partition
    parts = partition_level(L, split(np.asarray(L["coords"]), n_parts))
    assert sum(p.n_owned for p in parts) == L["nel"]

    whole = mgcfd.Solver.from_arrays([L], mg.mesh_variant)
    q0 = perturbed_state(L["nel"], whole.far_field()[:5], seed=seed)
    whole.set(0, "variables", q0)
    whole.set_option("fuse_update", 0)
    whole.smooth(0, sweeps)
    want_v, want_res, want_rms = whole.get(0, "variables"), whole.get(0, "residuals"), whole.calc_rms(0)
    whole.close()

    # one explicit stream for all three parts and the in-process copies (each thread makes it current):
    # the library's own streams are non-blocking and would not be ordered with torch's copies
    tstream = torch.cuda.Stream()
    stream = tstream.cuda_stream
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
            get = (lambda x: x.s.partial_min_tensor(0)) if partials else (lambda x: x.s.min_tensor(0))
            m = torch.stack([get(x) for x in sweepers]).min(dim=0).values
            for x in sweepers:
                get(x).copy_(m)
        barrier.wait()

    for P in parts:
        s = mgcfd.Solver.from_arrays([P.level], mg.mesh_variant, n_owned=[P.n_owned])
        s.set_option("flux_variant", variant)
        s.set_stream(stream)
        s.set(0, "variables", q0[P.global_ids])          # ghosts start current
        solvers.append(s)
        sweepers.append(PartitionedSweep(HipSolverAdapter(s, dev), P, None, exchange=exchange, allreduce_min=allreduce_min,
                                         make_buffer=lambda n: torch.empty(n, dtype=torch.float64, device=dev), fused=fused))
    errors = []

    def run(sw):
        try:
            torch.cuda.set_device(0)
            torch.cuda.set_stream(tstream)                # the current stream is per thread
            for _ in range(sweeps):
                sw.sweep()
        except Exception as e:                           # pragma: no cover
            errors.append(e)
            barrier.abort()

    threads = [threading.Thread(target=run, args=(sw,)) for sw in sweepers]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    sumsq = 0.0
    for P, s in zip(parts, solvers):
        own = P.global_ids[:P.n_owned]
        assert np.array_equal(s.get(0, "variables")[:P.n_owned].view(np.int64), want_v[own].view(np.int64))
        assert np.array_equal(s.get(0, "residuals")[:P.n_owned].view(np.int64), want_res[own].view(np.int64))
        # ghosts hold the owners' values after the last exchange
        assert np.array_equal(s.get(0, "variables")[P.n_owned:].view(np.int64), want_v[P.global_ids[P.n_owned:]].view(np.int64))
        sumsq += s.calc_rms(0) ** 2 * P.n_owned         # the library's RMS of a partitioned level is over its owned nodes
        s.close()
    assert abs(np.sqrt(sumsq / L["nel"]) - want_rms) <= 1e-12 * want_rms


@pytest.mark.parametrize("sizes,n_parts,fused", [((12, 6, 3), 3, False), ((14, 7), 4, False), ((12, 6, 3), 3, True),
                                                 ((3000, 700, 150), 3, True), ((2500, 500), 2, False)])   # node counts >= 100: tetrahedra
def test_partitioned_hierarchy_vcycles_equal_whole_mesh(sizes, n_parts, fused):
    """Every level of a hierarchy split over "ranks" (threads, one solver each, in-process copies for send/recv):
    flux ghosts, the children a rank's coarse nodes need for mgcfd_restrict and the parents mgcfd_prolong reads are
    ghosts that the cycle keeps current by halo exchanges; children are averaged in global-id order.  Three V-cycles
    must equal mgcfd_run_cycles on the whole mesh bit for bit on the owned nodes of every level."""
    import threading
    import torch
    import mgcfd
    from mgcfd import meshgen
    from mgcfd.distributed import HipSolverAdapter, PartitionedCycle
    from mgcfd.partition import partition_hierarchy, rcb_partition
    if sizes[0] >= 100:
        mg = meshgen.make_tet_multigrid(sizes, "m6wing", seed=4)
    else:
        mg = meshgen.make_multigrid(sizes, "m6wing", seed=4, jitter=0.2, area_noise=0.05, volume_noise=0.05)
    _partitioned_hierarchy_check(mg, n_parts, fused)


def _partitioned_hierarchy_check(mg, n_parts, fused, cycles=3):
    """The hierarchy `mg` split over n_parts solvers on this GPU against mgcfd_run_cycles on the whole (see the test above)."""
    import threading
    import torch
    import mgcfd
    from mgcfd.distributed import HipSolverAdapter, PartitionedCycle
    from mgcfd.partition import partition_hierarchy, rcb_partition
    dev = torch.device("cuda", 0)
    levels = mgcfd.generated_to_levels(mg)
    whole = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    whole.run_cycles(cycles)
    want = [whole.get(l, "variables") for l in range(len(levels))]
    want_res0 = whole.get(0, "residuals")
    whole.close()

    H = partition_hierarchy(levels, rcb_partition(np.asarray(levels[0]["coords"]), n_parts))
    assert all(sum(h.levels[l].n_owned for h in H) == levels[l]["nel"] for l in range(len(levels)))
    tstream = torch.cuda.Stream()
    solvers, cyclers = [], []
    barrier = threading.Barrier(n_parts)

    def exchange(cy, level):
        barrier.wait()
        for peer, buf in cy.buf_recv[level].items():
            buf.copy_(cyclers[peer].buf_send[level][cy.h.rank])
        barrier.wait()

    def allreduce_min(cy, level, partials=False):
        barrier.wait()
        if cy.h.rank == 0:
            get = (lambda c: c.s.partial_min_tensor(level)) if partials else (lambda c: c.s.min_tensor(level))
            m = torch.stack([get(c) for c in cyclers]).min(dim=0).values
            for c in cyclers:
                get(c).copy_(m)
        barrier.wait()

    for h in H:
        lv, owned, keys = h.solver_args()
        s = mgcfd.Solver.from_arrays(lv, mg.mesh_variant, n_owned=owned, order_keys=keys)
        s.set_stream(tstream.cuda_stream)
        solvers.append(s)
        cyclers.append(PartitionedCycle(HipSolverAdapter(s, dev), h, None, exchange=exchange, allreduce_min=allreduce_min,
                                        make_buffer=lambda n: torch.empty(max(n, 1), dtype=torch.float64, device=dev), fused=fused))
    errors = []

    def run(cy):
        try:
            torch.cuda.set_device(0)
            torch.cuda.set_stream(tstream)
            for _ in range(cycles):
                cy.cycle()
        except Exception as e:                               # pragma: no cover
            errors.append(e)
            barrier.abort()

    threads = [threading.Thread(target=run, args=(c,)) for c in cyclers]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for h, s in zip(H, solvers):
        for l, P in enumerate(h.levels):
            own = P.global_ids[:P.n_owned]
            got = s.get(l, "variables")[:P.n_owned]
            assert np.array_equal(got.view(np.int64), want[l][own].view(np.int64)), f"rank {h.rank} level {l}"
        own0 = h.levels[0].global_ids[:h.levels[0].n_owned]
        assert np.array_equal(s.get(0, "residuals")[:len(own0)].view(np.int64), want_res0[own0].view(np.int64))
        s.close()


def test_one_level_per_solver_equals_run_cycles(mesh3_dir):
    """BASELINE config 4 in miniature: the three levels of the case split over two solvers (two "ranks": levels 0
    and 2 on one, level 1 on the other), the restricted variables and the coarse residuals handed over as whole
    arrays through tensors that alias library memory (threads for ranks, a queue for RCCL send/recv).  Three
    cycles must equal mgcfd_run_cycles on one solver bit for bit, on every level."""
    import queue
    import threading
    import torch
    import mgcfd
    from mgcfd.distributed import HipSolverAdapter, LevelPerRankCycle
    dev = torch.device("cuda", 0)
    cycles, world = 3, 2
    whole = mgcfd.Solver.from_mesh(mgcfd.Mesh("input.dat", mesh3_dir))
    whole.run_cycles(cycles)
    want = [whole.get(l, "variables") for l in range(whole.num_levels)]
    n_levels = whole.num_levels
    whole.close()

    tstream = torch.cuda.Stream()
    solvers = [mgcfd.Solver.from_mesh(mgcfd.Mesh("input.dat", mesh3_dir)) for _ in range(world)]
    for s in solvers:
        s.set_stream(tstream.cuda_stream)
    boxes = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
    errors = []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            torch.cuda.set_stream(tstream)
            send = lambda t, dst: boxes[(rank, dst)].put(t.clone())
            recv = lambda t, src: t.copy_(boxes[(src, rank)].get(timeout=60))
            cyc = LevelPerRankCycle(HipSolverAdapter(solvers[rank], dev), n_levels, rank, world, send=send, recv=recv)
            for _ in range(cycles):
                cyc.cycle()
        except Exception as e:                               # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for l in range(n_levels):
        got = solvers[l % world].get(l, "variables")
        assert np.array_equal(got.view(np.int64), want[l].view(np.int64)), f"level {l}"
    for s in solvers:
        s.close()


@pytest.mark.parametrize("kind", ["random graph", "tetrahedra", "hub"])
def test_split_sweep_on_a_ragged_level_equals_smooth(kind):
    """The split sweep (mgcfd_sweep_begin / _flux0 / _end, both exchanged quantities) on levels whose tiles overflow
    the LDS halo (a random graph, Delaunay tetrahedra): the second stage cannot absorb the first stage's time_step
    there (role 5 needs every staged node in LDS) and the separate time_step launch is used; and on a 600-spoke hub,
    where it can, with the hub's long row on the workgroup's list — same bits as mgcfd_smooth every time."""
    import mgcfd
    from mgcfd import meshgen
    lvl = {"random graph": lambda: meshgen.make_random_graph_level(3000, degree=6, seed=5),
           "tetrahedra": lambda: meshgen.make_tet_level(4000, seed=5),
           "hub": lambda: meshgen.make_hub_level(600, scale=1e-4, seed=5)}[kind]()
    mg = meshgen.MultigridMesh(mesh_name="m6wing", levels=[lvl])
    L = mgcfd.generated_to_levels(mg)[0]
    ref = mgcfd.Solver.from_arrays([L], mg.mesh_variant)
    q = perturbed_state(L["nel"], ref.far_field()[:5], seed=31)
    ref.set(0, "variables", q)
    ref.smooth(0, 3)
    want = ref.get(0, "variables")
    ref.close()
    for partials in (True, False):
        s = mgcfd.Solver.from_arrays([L], mg.mesh_variant)
        s.set(0, "variables", q)
        for _ in range(3):
            (s.sweep_begin_partials if partials else s.sweep_begin)(0)
            s.sweep_flux0(0)
            (s.sweep_end_partials if partials else s.sweep_end)(0)
        assert np.array_equal(s.get(0, "variables").view(np.int64), want.view(np.int64)), partials
        s.close()


def test_rccl_accepts_the_aliased_tensors(mesh3_dir):
    """A one-rank RCCL group (backend "nccl"): the collectives the multi-rank paths issue — all-reduce(MIN) on the
    time-step scalar (async, as ShardedSweep does around sweep_flux0), all-reduce(SUM) on the RMS scalar and a
    broadcast on a whole level array — run on tensors that alias library memory and leave the values intact."""
    import socket
    import torch
    import torch.distributed as dist
    import mgcfd
    from mgcfd.distributed import HipSolverAdapter
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        st = torch.cuda.Stream()
        torch.cuda.set_stream(st)
        s = mgcfd.Solver.from_mesh(mgcfd.Mesh("input.dat", mesh3_dir))
        s.set_stream(st.cuda_stream)
        s.set(0, "variables", perturbed_state(s.nel(0), s.far_field()[:5], seed=3))
        ad = HipSolverAdapter(s, torch.device("cuda", 0))
        ref = mgcfd.Solver.from_mesh(mgcfd.Mesh("input.dat", mesh3_dir))
        ref.set(0, "variables", perturbed_state(s.nel(0), s.far_field()[:5], seed=3))
        ref.smooth(0, 2)
        s.sweep_begin(0)
        work = dist.all_reduce(ad.min_tensor(0), op=dist.ReduceOp.MIN, async_op=True)
        s.sweep_flux0(0)
        work.wait()
        s.sweep_end(0)
        s.sweep_begin_partials(0)                            # the partial minima as the exchanged quantity
        assert ad.partial_min_tensor(0).numel() == (s.nel(0) + 255) // 256
        work = dist.all_reduce(ad.partial_min_tensor(0), op=dist.ReduceOp.MIN, async_op=True)
        s.sweep_flux0(0)
        work.wait()
        s.sweep_end_partials(0)
        t = ad.sumsq_tensor(0).clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dist.broadcast(ad.array_tensor(0, "variables"), src=0)
        assert np.array_equal(s.get(0, "variables").view(np.int64), ref.get(0, "variables").view(np.int64))
        assert abs(np.sqrt(float(t.item()) / s.nel(0)) - ref.calc_rms(0)) <= 1e-12 * ref.calc_rms(0)
        s.close(); ref.close()
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream())
        dist.destroy_process_group()


def test_min_scalar_aliases_device_memory(mesh3_dir):
    """The all-reduce acts on a torch tensor that must alias the library's device scalar."""
    import torch
    import mgcfd
    from mgcfd.distributed import HipSolverAdapter
    s = mgcfd.Solver.from_mesh(mgcfd.Mesh("input.dat", mesh3_dir))
    with pytest.raises(ValueError):
        s.set_stream(0)                                   # the legacy default stream cannot be shared
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    s.set_stream(st.cuda_stream)
    s.set(0, "variables", perturbed_state(s.nel(0), s.far_field()[:5], seed=3))
    ad = HipSolverAdapter(s, torch.device("cuda", 0))
    s.step_factor_local(0)
    t = ad.min_tensor(0)
    assert t.dtype == torch.float64 and t.is_cuda and t.numel() == 1
    m = float(t.item())
    s.step_factor_apply(0)
    vol = s.get(0, "volumes")
    assert np.array_equal(s.get(0, "step_factors"), m / vol)
    t.fill_(0.25 * m)                      # what an all-reduce(MIN) with a smaller remote value does
    s.step_factor_apply(0)
    assert np.array_equal(s.get(0, "step_factors"), (0.25 * m) / vol)
    torch.cuda.set_stream(torch.cuda.default_stream())
    s.close()


# ------------------------------------------------------------------------------------------
# BASELINE-size level (67^3 = 300,763 nodes / 888,822 edges): size-independent properties
# ------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def big():
    import bench
    import mgcfd
    mg, levels = bench.build_workload(67)
    s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    q = bench.perturbed_state(s.nel(0), s.far_field()[:5])
    yield mgcfd, s, q, levels
    s.close()


def test_full_size_internal_flux_is_conservative(big):
    """Every internal edge adds F to one end and -F to the other (flux_kernel.elemfunc.c:142-189),
    so the internal fluxes of all nodes sum to zero up to rounding."""
    mgcfd, s, q, levels = big
    s.set(0, "variables", q)
    s.zero_fluxes(0)
    s.compute_flux_edge(0)
    f = s.get(0, "fluxes")
    assert np.isfinite(f).all()
    assert np.all(np.abs(f.sum(axis=0)) <= 1e-12 * np.abs(f).sum(axis=0))
    assert s.loop_iters(0)["flux"] % levels[0]["n_internal"] == 0


def test_full_size_flux_variants_agree(big):
    """On the 300K-node level every flux variant writes the same bits (one launch, all edge classes)."""
    mgcfd, s, q, levels = big
    assert s.has_edge_once(0)
    out = {}
    assert s.has_half_rows(0)
    for v in (0, 1, 2, 3, 4, 16, 32):
        s.set_option("flux_variant", v)
        s.set(0, "variables", q)
        s.zero_fluxes(0)
        s.compute_fluxes(0)
        out[v] = s.get(0, "fluxes")
    s.set_option("flux_variant", 0)
    for v in (1, 2, 3, 4, 16, 32):
        assert np.array_equal(out[0].view(np.int64), out[v].view(np.int64)), v


def test_full_size_deterministic_and_modes_agree(big):
    mgcfd, s, q, levels = big
    runs = {}
    for name, opts in {"fused": dict(fuse_update=1, graph=1), "fused-again": dict(fuse_update=1, graph=1),
                       "eager": dict(fuse_update=1, graph=0), "unfused": dict(fuse_update=0, graph=0),
                       "fast": dict(fuse_update=1, graph=1, exact=0)}.items():
        s.set_option("exact", opts.get("exact", 1))
        s.set_option("fuse_update", opts["fuse_update"])
        s.set_option("graph", opts["graph"])
        s.set(0, "variables", q)
        s.zero_fluxes(0)
        s.smooth(0, 4)
        runs[name] = (s.get(0, "variables"), s.get(0, "residuals"))
        assert s.check_for_invalid_variables(0)[0] == 0
    s.set_option("exact", 1)
    for other in ("fused-again", "eager", "unfused"):          # no atomics, fixed order: bit-reproducible
        for a, b in zip(runs["fused"], runs[other]):
            assert np.array_equal(a.view(np.int64), b.view(np.int64)), other
    for a, b in zip(runs["fused"], runs["fast"]):                # FMA contraction only
        assert np.abs(a - b).max() <= REL_FAST * np.abs(a).max()
    assert not np.array_equal(runs["fused"][0], q)               # the sweeps did change the state


def test_full_size_matches_oracle_on_one_sweep(big, oracle):
    """One whole sweep on the 300K-node level against the oracle (about a second of CPU)."""
    import ctypes as C
    mgcfd, s, q, levels = big
    L = levels[0]
    lib = oracle.load()
    edges = np.ascontiguousarray(L["edges"]).copy()
    coords = np.ascontiguousarray(L["coords"], dtype=np.float64)
    lib.ora_adjust_ewt(oracle.ptr(coords), len(edges), oracle.ptr(edges))
    lib.ora_dampen_ewt(len(edges), oracle.ptr(edges), 5e-8)
    ff = oracle.farfield()
    nel, ni, nb, nw = L["nel"], L["n_internal"], L["n_boundary"], L["n_wall"]
    v, old, f, sf = q.copy(), q.copy(), np.zeros_like(q), np.zeros(nel)
    vol = np.ascontiguousarray(L["volumes"], dtype=np.float64)
    lib.ora_compute_step_factor(nel, oracle.ptr(v), oracle.ptr(vol), oracle.ptr(sf))
    for j in range(3):
        lib.ora_compute_flux_edge(0, ni, oracle.ptr(edges), oracle.ptr(v), oracle.ptr(f))
        lib.ora_compute_boundary_flux_edge(ni, nb, oracle.ptr(edges), oracle.ptr(v), oracle.ptr(f))
        lib.ora_compute_wall_flux_edge(ni + nb, nw, oracle.ptr(edges), oracle.ptr(v), oracle.ptr(f), C.byref(ff))
        lib.ora_time_step(j, nel, oracle.ptr(sf), oracle.ptr(f), oracle.ptr(old), oracle.ptr(v))
    s.set_option("exact", 1)
    s.set_option("fuse_update", 1)
    s.set_option("graph", 1)
    s.set(0, "variables", q)
    s.zero_fluxes(0)
    s.smooth(0, 1)
    _assert_close(s.get(0, "variables"), v, True, "300K-node sweep")
    _assert_close(s.get(0, "step_factors"), sf, True, "300K-node step factors")


# ------------------------------------------------------------------------------------------
# An unstructured level of the same edge count (Delaunay tetrahedra, 120,000 nodes / 926,505 edges):
# size-independent properties, and the oracle on one sweep
# ------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def big_tet():
    import mgcfd
    from mgcfd import meshgen
    # fvcorr: the weights are used as they are (the m6wing / rotor37 adjust_ewt divides internal weights by the edge
    # length, after which the dual cells no longer close); every hull face far field
    mg = meshgen.MultigridMesh(mesh_name="fvcorr")
    mg.levels.append(meshgen.make_tet_level(120000, seed=0, wall_below=2.0))
    levels = mgcfd.generated_to_levels(mg)
    s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    yield mgcfd, s, levels, mg
    s.close()


def test_tetrahedral_level_preserves_the_free_stream_and_conserves(big_tet):
    """Closed median-dual cells: the far-field state is a steady solution (fluxes vanish to rounding, sweeps leave it
    where it is), and on a perturbed state the internal fluxes of all nodes cancel (every edge adds F and -F) — with
    the long rows of this mesh on the workgroups' lists."""
    mgcfd, s, levels, mg = big_tet
    t = s.tiling(0)
    assert t["list_entries"] > 0 and t["coordinate_boxes"] == 1
    ff = s.far_field()[:5]
    q = np.tile(ff, (s.nel(0), 1))
    s.set(0, "variables", q)
    s.zero_fluxes(0)
    s.compute_flux_edge(0); s.compute_boundary_flux_edge(0); s.compute_wall_flux_edge(0)
    f = s.get(0, "fluxes")
    w = np.abs(levels[0]["edges"]["x"]).max()
    assert np.abs(f).max() < 1e-12 * w * np.abs(ff).max() * 40             # ~40 faces per node, each O(w * |q|)
    s.zero_fluxes(0)
    s.smooth(0, 5)
    assert np.abs(s.get(0, "variables") - q).max() < 1e-9                  # (step factors ~1e3: rounding residue, not drift)
    from conftest import perturbed_state
    s.set(0, "variables", perturbed_state(s.nel(0), ff, seed=4))
    s.zero_fluxes(0)
    s.compute_flux_edge(0)
    f = s.get(0, "fluxes")
    assert np.isfinite(f).all() and np.all(np.abs(f.sum(axis=0)) <= 1e-12 * np.abs(f).sum(axis=0))


def test_tetrahedral_level_matches_oracle_on_one_sweep(big_tet, oracle):
    """One sweep on the 120,000-node tetrahedral level against the oracle, bit for bit (the oracle needs ~0.3 s)."""
    mgcfd, s, levels, mg = big_tet
    from conftest import perturbed_state
    q = perturbed_state(s.nel(0), s.far_field()[:5], seed=8)
    lib = oracle.load()
    L = levels[0]
    edges = np.ascontiguousarray(L["edges"]).copy()                      # (fvcorr: no adjust_ewt / dampen_ewt)
    ni, nb, nw = L["n_internal"], L["n_boundary"], L["n_wall"]
    nel = L["nel"]
    vol = np.ascontiguousarray(L["volumes"], dtype=np.float64)
    var, old, flux, sf = q.copy(), q.copy(), np.zeros((nel, 5)), np.zeros(nel)
    ff = oracle.farfield()
    lib.ora_compute_step_factor_legacy(nel, oracle.ptr(var), oracle.ptr(vol), oracle.ptr(sf))
    for j in range(3):
        lib.ora_compute_flux_edge(0, ni, oracle.ptr(edges), oracle.ptr(var), oracle.ptr(flux))
        lib.ora_compute_boundary_flux_edge(ni, nb, oracle.ptr(edges), oracle.ptr(var), oracle.ptr(flux))
        lib.ora_compute_wall_flux_edge(ni + nb, nw, oracle.ptr(edges), oracle.ptr(var), oracle.ptr(flux), C.byref(ff))
        lib.ora_time_step(j, nel, oracle.ptr(sf), oracle.ptr(flux), oracle.ptr(old), oracle.ptr(var))
    s.set(0, "variables", q)
    s.zero_fluxes(0)
    s.smooth(0, 1)
    assert np.array_equal(s.get(0, "variables").view(np.int64), var.view(np.int64))



@pytest.mark.parametrize("case,gpus", [("fvcorr_1lvl", 3), ("m6_3lvl", 2), ("m6_3lvl", 3), ("m6_2lvl_dup2", 2)])
def test_driver_on_several_gpus_reproduces_reference_binary(case, gpus, tmp_path):
    """euler3d_gpu_double --gpus N (all ranks on this one GPU: --gpus-share-device): a single-level input partitioned
    over the ranks with halo messages after every stage, a multigrid input with one level per rank — the variables
    dump must still be the reference binary's byte for byte, the RMS lines and the loop counters the same."""
    d, cycles, dup = _case(case)
    exe = os.path.join(ROOT, "mg-cfd-app-plain_amd", "csrc", "euler3d_gpu_double")
    cmd = [exe, "-i", "input.dat", "-d", os.path.join(d, "input"), "-o", str(tmp_path) + "/", "-g", str(cycles),
           "-m", str(dup), "--output-variables", "--gpus", str(gpus), "--gpus-share-device"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    dump = tmp_path / f"variables.size={dup}x.cycles={cycles}.level=0"
    assert dump.read_bytes() == open(os.path.join(d, "variables.level0.txt"), "rb").read()
    want_lines = [l.strip() for l in open(os.path.join(d, "stdout.txt")) if "RMS" in l]
    got_lines = [l.strip() for l in r.stdout.splitlines() if "RMS" in l]
    assert got_lines == want_lines
    want, got = _csv_row(os.path.join(d, "LoopNumIters.csv")), _csv_row(tmp_path / "LoopNumIters.csv")
    assert list(got.keys()) == list(want.keys())
    for k in want:
        if k[:-1] in ("flux", "update", "compute_step", "time_step", "restrict", "prolong") or k in ("Size", "Mesh", "MG cycles"):
            assert got[k] == want[k], k
    assert got["Num threads"] == str(min(gpus, 3 if case == "m6_3lvl" else (2 if case == "m6_2lvl_dup2" else gpus)))
    assert f"{int(got['Num threads'])} ranks" in r.stderr


@pytest.mark.parametrize("case,gpus", [("fvcorr_1lvl", 3), ("m6_3lvl", 2)])
def test_driver_on_several_gpus_validates_and_dumps_as_on_one(case, gpus, tmp_path):
    """--gpus N honours -v and every dump flag (the one-GPU path and the reference do, src/euler3d_cpu_double.cpp:704-772):
    PASS against the reference binary's own dump as the solution file, exit 1 with the reference's message against a
    spoiled one, and variables / fluxes / step-factor dumps equal to the one-GPU run's byte for byte."""
    import shutil
    d, cycles, dup = _case(case)
    work = tmp_path / "in"
    shutil.copytree(os.path.join(d, "input"), work)
    sol = work / f"solution.variables.size={dup}x.cycles={cycles}.level=0"
    shutil.copy(os.path.join(d, "variables.level0.txt"), sol)
    exe = os.path.join(ROOT, "mg-cfd-app-plain_amd", "csrc", "euler3d_gpu_double")
    outs = {}
    for name, extra in (("one", []), ("many", ["--gpus", str(gpus), "--gpus-share-device"])):
        out = tmp_path / name
        out.mkdir()
        r = subprocess.run([exe, "-i", "input.dat", "-d", str(work), "-o", str(out) + "/", "-g", str(cycles), "-m", str(dup), "-v",
                            "--output-variables", "--output-fluxes", "--output-step-factors"] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "NaN check passed" in r.stdout and "PASS: variables[] validated successfully" in r.stdout, r.stdout
        outs[name] = out
    for arr in ("variables", "fluxes", "step_factors"):
        f = f"{arr}.size={dup}x.cycles={cycles}.level=0"
        assert (outs["many"] / f).read_bytes() == (outs["one"] / f).read_bytes(), arr
    vals = np.loadtxt(sol)
    vals[7, 1] += 1e-6
    np.savetxt(sol, vals, fmt="%.17e")
    r = subprocess.run([exe, "-i", "input.dat", "-d", str(work), "-o", str(tmp_path) + "/", "-g", str(cycles), "-m", str(dup), "-v",
                        "--gpus", str(gpus), "--gpus-share-device"], capture_output=True, text=True)
    assert r.returncode != 0 and "ERROR: Unacceptable error detected at (i=7, v=1)" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("case,gpus,flag", [("m6_3lvl", 4, []), ("m6_3lvl", 2, ["--gpus-partition"]), ("m6_2lvl_dup2", 3, []), ("tet_2lvl", 3, [])])
def test_driver_on_a_partitioned_hierarchy_reproduces_reference_binary(case, gpus, flag, tmp_path):
    """euler3d_gpu_double --gpus N with N above the number of levels (or --gpus-partition): EVERY level split over the N ranks
    (multi_gpu.cpp: partition_hierarchy), the V-cycles swept inside the library (mgcfd_group_cycles).  The variables dump
    must still be the reference binary's byte for byte, the RMS lines and the loop counters the same; -v passes."""
    import shutil
    d, cycles, dup = _case(case)
    work = tmp_path / "in"
    shutil.copytree(os.path.join(d, "input"), work)
    shutil.copy(os.path.join(d, "variables.level0.txt"), work / f"solution.variables.size={dup}x.cycles={cycles}.level=0")
    exe = os.path.join(ROOT, "mg-cfd-app-plain_amd", "csrc", "euler3d_gpu_double")
    cmd = [exe, "-i", "input.dat", "-d", str(work), "-o", str(tmp_path) + "/", "-g", str(cycles),
           "-m", str(dup), "--output-variables", "-v", "--gpus", str(gpus), "--gpus-share-device"] + flag
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "every level partitioned" in r.stderr and f"{gpus} ranks" in r.stderr
    dump = tmp_path / f"variables.size={dup}x.cycles={cycles}.level=0"
    assert dump.read_bytes() == open(os.path.join(d, "variables.level0.txt"), "rb").read()
    assert "PASS: variables[] validated successfully" in r.stdout
    want_lines = [l.strip() for l in open(os.path.join(d, "stdout.txt")) if "RMS" in l]
    got_lines = [l.strip() for l in r.stdout.splitlines() if "RMS" in l]
    assert got_lines == want_lines
    want, got = _csv_row(os.path.join(d, "LoopNumIters.csv")), _csv_row(tmp_path / "LoopNumIters.csv")
    for k in want:
        if k[:-1] in ("flux", "update", "compute_step", "time_step", "restrict", "prolong") or k in ("Size", "Mesh", "MG cycles"):
            assert got[k] == want[k], k
    assert got["Num threads"] == str(gpus)


def test_driver_refuses_more_gpus_than_there_are(tmp_path):
    d, cycles, dup = _case("fvcorr_1lvl")
    exe = os.path.join(ROOT, "mg-cfd-app-plain_amd", "csrc", "euler3d_gpu_double")
    r = subprocess.run([exe, "-i", "input.dat", "-d", os.path.join(d, "input"), "-o", str(tmp_path) + "/", "-g", "1", "--gpus", "64"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "GPU(s) are visible" in r.stderr
