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


def _run_both(mgcfd, oracle, directory, cycles, exact, indirect_rw=False, duplicate=1, fuse=True):
    mesh = mgcfd.Mesh("input.dat", directory, duplicate)
    solver = mgcfd.Solver.from_mesh(mesh)
    solver.set_option("exact", int(exact))
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


def test_vcycles_with_indirect_rw_and_duplication(oracle, mesh_dir):
    import mgcfd
    _run_both(mgcfd, oracle, mesh_dir, 3, True, indirect_rw=True, duplicate=2)


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("exact", [True, False])
def test_fvcorr_single_level(oracle, fvcorr_dir, exact, fuse):
    import mgcfd
    _run_both(mgcfd, oracle, fvcorr_dir, 50, exact, fuse=fuse)


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
