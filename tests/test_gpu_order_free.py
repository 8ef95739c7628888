"""The ORDER-FREE flux kernel (k_flux_free: `fast` namespace, MGCFD_OPT_EXACT = 0 with MGCFD_OPT_FLUX_VARIANT bit 6) against
the oracle.  It evaluates every internal edge of a tile once and hands the other end its share through LDS fp64 atomics,
so its sums are associated differently from the reference's and differ from run to run in the last bits.  The bar
(north_star: "residuals within 1e-10 of CPU reference"), written here:

    REL_LAUNCH = 1e-12   one launch / one sweep, max |difference| / max |reference value| per array
    REL_RUN    = 1e-10   level-0 `variables` after 25 full-size V-cycles
    the reference's own -v rule (validation.cpp:140-199) passes on that state

The bit-identical kernels stay the default and the parity gate (tests/test_gpu_parity.py, tests/test_gpu_configs.py)."""
import ctypes as C

import numpy as np
import pytest

from conftest import perturbed_state
from test_gpu_configs import _oracle_levels

pytestmark = pytest.mark.gpu

REL_LAUNCH = 1e-12
REL_RUN = 1e-10
FREE = 64 | 1          # order-free accumulation, edge-length factor recomputed


def _rel(got, want):
    return np.abs(got - want).max() / max(np.abs(want).max(), 1e-300)


def test_order_free_flux_kernels_against_the_oracle(oracle, mesh3_dir):
    """Every edge class on its own onto a non-zero flux array ('+=' launches), and the three classes in one launch from
    zero, on the three levels of a small hierarchy (tiles with cavities, boundary and far-field faces)."""
    import mgcfd
    mesh = mgcfd.Mesh("input.dat", mesh3_dir)
    s = mgcfd.Solver.from_mesh(mesh)
    s.set_option("exact", 0)
    s.set_option("flux_variant", FREE)
    case = oracle.OracleCase.from_input_dat(mesh3_dir + "/input.dat")
    lib = oracle.load()
    ff = oracle.farfield()
    ran_free = 0
    for l in range(case.nlevels):
        L = case.levels[l]
        lib.ora_adjust_ewt(L.coords, L.n_edges, L.edges)
        lib.ora_dampen_ewt(L.n_edges, L.edges, 5e-8)
        ran_free += int(s.has_order_free(l))
        q = perturbed_state(L.nel, ff.var, seed=900 + l)
        f0 = np.random.default_rng(17 + l).normal(size=(L.nel, 5)) * 1e-7
        steps = [(lambda f: lib.ora_compute_flux_edge(L.internal_start, L.n_internal, L.edges, oracle.ptr(q), oracle.ptr(f)), s.compute_flux_edge),
                 (lambda f: lib.ora_compute_boundary_flux_edge(L.boundary_start, L.n_boundary, L.edges, oracle.ptr(q), oracle.ptr(f)), s.compute_boundary_flux_edge),
                 (lambda f: lib.ora_compute_wall_flux_edge(L.wall_start, L.n_wall, L.edges, oracle.ptr(q), oracle.ptr(f), C.byref(ff)), s.compute_wall_flux_edge)]
        s.set(l, "variables", q)
        want = f0.copy()
        s.set(l, "fluxes", f0)
        for ora_fn, gpu_fn in steps:
            ora_fn(want)
            gpu_fn(l)
            assert _rel(s.get(l, "fluxes"), want) <= REL_LAUNCH, f"level {l}: accumulating launch"
        s.zero_fluxes(l)
        s.compute_fluxes(l)
        want = np.zeros((L.nel, 5))
        for ora_fn, _ in steps:
            ora_fn(want)
        assert _rel(s.get(l, "fluxes"), want) <= REL_LAUNCH, f"level {l}: all classes from zero"
    assert ran_free >= 1, "no level of this hierarchy has a half-row plan: the order-free kernel never ran"
    s.close()


def test_order_free_sweeps_and_cycles_on_small_hierarchies(oracle, mesh_dir, mesh3_dir, fvcorr_dir):
    """Whole cycles (fused stages, transfers) with the order-free stages, global and local time step: every level's state
    against ora_solve after a few cycles."""
    import mgcfd
    for d, cycles in ((mesh_dir, 6), (mesh3_dir, 4), (fvcorr_dir, 10)):
        s = mgcfd.Solver.from_mesh(mgcfd.Mesh("input.dat", d))
        s.set_option("exact", 0)
        s.set_option("flux_variant", FREE)
        rms = s.run_cycles(cycles)
        case = oracle.OracleCase.from_input_dat(d + "/input.dat")
        rc, want_rms, _ = case.solve(cycles)
        assert rc == 0
        for l in range(s.num_levels):
            assert _rel(s.get(l, "variables"), case.array(l, "variables").reshape(-1, 5)) <= REL_RUN, f"{d}: level {l}"
        assert np.allclose(rms, want_rms, rtol=1e-9, atol=0)
        s.close()


def test_order_free_is_off_in_the_exact_namespace_and_where_no_plan_exists(oracle):
    """Bit 6 means nothing to the bit-identical kernels; and a level without a half-row plan (a degree-10 random graph:
    long rows) runs the contracted node gather, whose sums keep the reference's order."""
    import mgcfd
    from mgcfd import meshgen
    mg = meshgen.make_multigrid((11,), "m6wing", seed=2, jitter=0.2, area_noise=0.05, volume_noise=0.05)
    levels = mgcfd.generated_to_levels(mg)
    q = perturbed_state(levels[0]["nel"], oracle.farfield().var, seed=5)
    out = {}
    for name, exact, v in (("exact", 1, 1), ("exact+bit6", 1, FREE)):
        s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
        s.set_option("exact", exact)
        s.set_option("flux_variant", v)
        s.set(0, "variables", q)
        s.smooth(0, 2)
        out[name] = s.get(0, "variables").copy()
        s.close()
    assert np.array_equal(out["exact"].view(np.int64), out["exact+bit6"].view(np.int64))
    rg = meshgen.MultigridMesh(mesh_name="m6wing")
    rg.levels.append(meshgen.make_random_graph_level(3000, degree=10, seed=4))
    lv = mgcfd.generated_to_levels(rg)
    q = perturbed_state(lv[0]["nel"], oracle.farfield().var, seed=6)
    res = {}
    for v in (1, FREE):
        s = mgcfd.Solver.from_arrays(lv, rg.mesh_variant)
        s.set_option("exact", 0)
        s.set_option("flux_variant", v)
        assert not s.has_order_free(0)                       # (tiles with halo nodes beyond the LDS image)
        s.set(0, "variables", q)
        s.compute_fluxes(0)
        res[v] = s.get(0, "fluxes").copy()
        s.close()
    assert np.array_equal(res[1].view(np.int64), res[FREE].view(np.int64))


def test_order_free_at_full_size(oracle):
    """BASELINE configs[1] and [2] at size.  One launch and one sweep on the 300,763-node level against the oracle's loops
    (<= 1e-12); then 25 V-cycles on the (67, 55, 48, 43)^3 hierarchy against ora_solve: level-0 variables <= 1e-10 and the
    reference's -v rule."""
    import bench
    import mgcfd
    lib = oracle.load()
    ff = oracle.farfield()
    mg, levels = bench.build_workload(67)
    L = levels[0]
    edges = np.ascontiguousarray(L["edges"]).copy()
    coords = np.ascontiguousarray(L["coords"], dtype=np.float64)
    lib.ora_adjust_ewt(oracle.ptr(coords), len(edges), oracle.ptr(edges))
    lib.ora_dampen_ewt(len(edges), oracle.ptr(edges), 5e-8)
    nel, ni, nb, nw = L["nel"], L["n_internal"], L["n_boundary"], L["n_wall"]
    vol = np.ascontiguousarray(L["volumes"], dtype=np.float64)
    s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    s.set_option("exact", 0)
    s.set_option("flux_variant", FREE)
    assert s.has_order_free(0)
    q = bench.perturbed_state(nel, s.far_field()[:5])
    s.set(0, "variables", q)
    s.zero_fluxes(0)
    s.compute_flux_edge(0)
    f = np.zeros_like(q)
    lib.ora_compute_flux_edge(0, ni, oracle.ptr(edges), oracle.ptr(q), oracle.ptr(f))
    assert _rel(s.get(0, "fluxes"), f) <= REL_LAUNCH, "compute_flux_edge at size"
    s.zero_fluxes(0)
    v, sf, old = q.copy(), np.zeros(nel), q.copy()
    f[:] = 0.0
    lib.ora_compute_step_factor(nel, oracle.ptr(v), oracle.ptr(vol), oracle.ptr(sf))
    for j in range(3):
        lib.ora_compute_flux_edge(0, ni, oracle.ptr(edges), oracle.ptr(v), oracle.ptr(f))
        lib.ora_compute_boundary_flux_edge(ni, nb, oracle.ptr(edges), oracle.ptr(v), oracle.ptr(f))
        lib.ora_compute_wall_flux_edge(ni + nb, nw, oracle.ptr(edges), oracle.ptr(v), oracle.ptr(f), C.byref(ff))
        lib.ora_time_step(j, nel, oracle.ptr(sf), oracle.ptr(f), oracle.ptr(old), oracle.ptr(v))
    s.smooth(0, 1)
    assert _rel(s.get(0, "variables"), v) <= REL_LAUNCH, "one sweep at size: variables"
    assert _rel(s.get(0, "residuals"), v - old) <= 1e-9, "one sweep at size: residuals (differences of nearly equal states)"
    s.close()

    mg, levels = bench.build_hierarchy()
    cycles = 25
    lv, keep = _oracle_levels(oracle, levels)
    want_rms = np.zeros(cycles)
    iters = (oracle.OraIters * len(levels))()
    assert lib.ora_solve(lv, len(levels), mg.mesh_variant, cycles, 0, oracle.ptr(want_rms), iters) == 0
    s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    s.set_option("exact", 0)
    s.set_option("flux_variant", FREE)
    assert all(s.has_order_free(l) for l in range(4))
    rms = s.run_cycles(cycles)
    got = s.get(0, "variables")
    assert _rel(got, keep[0]["variables"]) <= REL_RUN, "level-0 variables after 25 V-cycles"
    assert lib.ora_identify_differences(oracle.ptr(np.ascontiguousarray(got)), oracle.ptr(keep[0]["variables"]), levels[0]["nel"], mg.mesh_variant) == -1
    assert np.allclose(rms, want_rms, rtol=1e-9, atol=0)
    s.close()


def test_order_free_on_levels_of_non_uniform_degree(oracle):
    """Slices with more than five half rows per lane (the loop behind the prologue's five): the mixed-element level at size
    (internal degrees 3 ... 14: 300,763 nodes / 1,004,901 edges) and a small mixed hierarchy through whole cycles, against the
    oracle's loops."""
    import bench
    import mgcfd
    from mgcfd import meshgen
    lib = oracle.load()
    ff = oracle.farfield()
    mg, levels = bench.build_workload(67, mesh="mixed")
    L = levels[0]
    edges = np.ascontiguousarray(L["edges"]).copy()
    coords = np.ascontiguousarray(L["coords"], dtype=np.float64)
    lib.ora_adjust_ewt(oracle.ptr(coords), len(edges), oracle.ptr(edges))
    lib.ora_dampen_ewt(len(edges), oracle.ptr(edges), 5e-8)
    nel, ni, nb, nw = L["nel"], L["n_internal"], L["n_boundary"], L["n_wall"]
    s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    s.set_option("exact", 0)
    s.set_option("flux_variant", FREE)
    assert s.has_order_free(0) and not s.has_half_rows(0)       # (more than five evaluations per lane somewhere: the ordered half-row kernel declines)
    q = bench.perturbed_state(nel, s.far_field()[:5])
    s.set(0, "variables", q)
    s.zero_fluxes(0)
    s.compute_fluxes(0)
    f = np.zeros_like(q)
    lib.ora_compute_flux_edge(0, ni, oracle.ptr(edges), oracle.ptr(q), oracle.ptr(f))
    lib.ora_compute_boundary_flux_edge(ni, nb, oracle.ptr(edges), oracle.ptr(q), oracle.ptr(f))
    lib.ora_compute_wall_flux_edge(ni + nb, nw, oracle.ptr(edges), oracle.ptr(q), oracle.ptr(f), C.byref(ff))
    assert _rel(s.get(0, "fluxes"), f) <= REL_LAUNCH
    s.close()
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        mgs = meshgen.make_mixed_multigrid((11, 6), "m6wing", seed=2, jitter=0.2, area_noise=0.05, volume_noise=0.05)
        meshgen.write_input(mgs, d)
        s = mgcfd.Solver.from_mesh(mgcfd.Mesh("input.dat", d))
        s.set_option("exact", 0)
        s.set_option("flux_variant", FREE)
        assert s.has_order_free(0)
        s.run_cycles(5)
        case = oracle.OracleCase.from_input_dat(d + "/input.dat")
        rc, _, _ = case.solve(5)
        assert rc == 0
        for l in range(2):
            assert _rel(s.get(l, "variables"), case.array(l, "variables").reshape(-1, 5)) <= REL_RUN
        s.close()


def test_order_free_on_tetrahedral_levels_with_halos_beyond_the_shared_table(oracle):
    """Delaunay tetrahedra with median-dual metrics (degrees 5 ... 50, halos of 300-500 nodes per tile: beyond the 303 ids of the
    table the node-gather kernels share): the order-free kernel stages from its own table, two halo nodes per thread, and
    spreads a high-degree node's evaluations over its tile's lanes.  One launch on a 30,000-node level and whole cycles on a
    tetrahedral hierarchy against the oracle."""
    import mgcfd
    from mgcfd import meshgen
    lib = oracle.load()
    ff = oracle.farfield()
    mg = meshgen.MultigridMesh(mesh_name="m6wing")
    mg.levels.append(meshgen.make_tet_level(30000, seed=1))
    levels = mgcfd.generated_to_levels(mg)
    L = levels[0]
    edges = np.ascontiguousarray(L["edges"]).copy()
    coords = np.ascontiguousarray(L["coords"], dtype=np.float64)
    lib.ora_adjust_ewt(oracle.ptr(coords), len(edges), oracle.ptr(edges))
    lib.ora_dampen_ewt(len(edges), oracle.ptr(edges), 5e-8)
    nel, ni, nb, nw = L["nel"], L["n_internal"], L["n_boundary"], L["n_wall"]
    s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    til = s.tiling(0)
    assert til["halo_max"] > til["halo_capacity"], "this level was meant to have tiles with more halo nodes than the shared table stages"
    assert s.has_order_free(0) and not s.has_half_rows(0)
    s.set_option("exact", 0)
    s.set_option("flux_variant", FREE)
    q = perturbed_state(nel, ff.var, seed=77)
    s.set(0, "variables", q)
    s.zero_fluxes(0)
    s.compute_fluxes(0)
    f = np.zeros_like(q)
    lib.ora_compute_flux_edge(0, ni, oracle.ptr(edges), oracle.ptr(q), oracle.ptr(f))
    lib.ora_compute_boundary_flux_edge(ni, nb, oracle.ptr(edges), oracle.ptr(q), oracle.ptr(f))
    lib.ora_compute_wall_flux_edge(ni + nb, nw, oracle.ptr(edges), oracle.ptr(q), oracle.ptr(f), C.byref(ff))
    assert _rel(s.get(0, "fluxes"), f) <= REL_LAUNCH
    s.close()
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        mgs = meshgen.make_tet_multigrid((6000, 1200), "m6wing", seed=3)
        meshgen.write_input(mgs, d)
        s = mgcfd.Solver.from_mesh(mgcfd.Mesh("input.dat", d))
        s.set_option("exact", 0)
        s.set_option("flux_variant", FREE)
        s.run_cycles(4)
        case = oracle.OracleCase.from_input_dat(d + "/input.dat")
        rc, _, _ = case.solve(4)
        assert rc == 0
        for l in range(2):
            assert _rel(s.get(l, "variables"), case.array(l, "variables").reshape(-1, 5)) <= REL_RUN
        s.close()
