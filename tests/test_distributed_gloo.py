"""Multi-process host logic on CPU: world_size 2 over gloo.

The N>1 path shards the problem as one mesh copy per rank, coupled only through the global
minimum time step (one all-reduce MIN per sweep) and the RMS (one all-reduce SUM) — exactly the
coupling the reference's `-m` duplication has.  Here the per-rank solver is a CPU stand-in
built on the ORACLE (test infrastructure; the product passes the HIP solver to the very same
mgcfd.distributed.ShardedSweep), and the result must equal a single-process oracle run on the
2x duplicated mesh."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_INPUT = os.path.join(ROOT, "tests", "golden", "m6_2lvl", "input", "input.dat")


class OracleRankSolver:
    """Kernel-granular stand-in for mgcfd.api.Solver backed by the C oracle, level 0 only."""

    def __init__(self, oracle, seed):
        import ctypes as C
        self.O, self.C = oracle, C
        self.lib = oracle.load()
        self.case = oracle.OracleCase.from_input_dat(GOLDEN_INPUT)
        self.L = self.case.levels[0]
        self.lib.ora_adjust_ewt(self.L.coords, self.L.n_edges, self.L.edges)
        self.lib.ora_dampen_ewt(self.L.n_edges, self.L.edges, 5e-8)
        self.ff = oracle.farfield()
        rng = np.random.default_rng(seed)
        q = np.tile(np.array(self.ff.var), (self.L.nel, 1)) * (1 + 0.02 * rng.uniform(-1, 1, (self.L.nel, 5)))
        self.case.array(0, "variables")[:] = q.ravel()
        self._min = torch.zeros(1, dtype=torch.float64)
        self._sumsq = torch.zeros(1, dtype=torch.float64)

    def a(self, name):
        return self.case.array(0, name)

    def copy_old_variables(self, l):
        self.a("old_variables")[:] = self.a("variables")

    def step_factor_local(self, l):
        # first half of compute_step_factor (cfd_loops.cpp:98-125): 0.5*cbrt(vol)/(|v|+c) and the LOCAL minimum
        q = self.a("variables").reshape(-1, 5)
        v = q[:, 1:4] / q[:, :1]
        sp2 = (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2]
        p = (1.4 - 1.0) * (q[:, 4] - 0.5 * q[:, 0] * sp2)
        c = np.sqrt(1.4 * p / q[:, 0])
        sf = 0.5 * (np.cbrt(self.a("volumes")) / (np.sqrt(sp2) + c))
        self._min[0] = float(sf.min())

    def min_tensor(self, l):
        return self._min

    def step_factor_apply(self, l):
        self.a("step_factors")[:] = float(self._min[0]) / self.a("volumes")

    def compute_fluxes(self, l):
        L, O = self.L, self.O
        q, f = self.a("variables"), self.a("fluxes")
        self.lib.ora_compute_flux_edge(L.internal_start, L.n_internal, L.edges, O.ptr(q), O.ptr(f))
        self.lib.ora_compute_boundary_flux_edge(L.boundary_start, L.n_boundary, L.edges, O.ptr(q), O.ptr(f))
        self.lib.ora_compute_wall_flux_edge(L.wall_start, L.n_wall, L.edges, O.ptr(q), O.ptr(f), self.C.byref(self.ff))

    def time_step(self, l, j):
        O = self.O
        self.lib.ora_time_step(j, self.L.nel, O.ptr(self.a("step_factors")), O.ptr(self.a("fluxes")),
                               O.ptr(self.a("old_variables")), O.ptr(self.a("variables")))

    def residual(self, l):
        self.a("residuals")[:] = self.a("variables") - self.a("old_variables")

    def sumsq_tensor(self, l):
        self._sumsq[0] = float((self.a("residuals") ** 2).sum())
        return self._sumsq


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, sweeps, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    from mgcfd.distributed import ShardedSweep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    solver = OracleRankSolver(oracle_py, seed=500 + rank)
    sw = ShardedSweep(solver, dist, global_time_step=True, fused=False)
    assert sw.world == world
    rms = []
    for _ in range(sweeps):
        sw.sweep(0)
        rms.append(sw.rms(0, solver.L.nel))
    np.save(os.path.join(out_dir, f"vars_{rank}.npy"), solver.a("variables").copy())
    np.save(os.path.join(out_dir, f"sf_{rank}.npy"), solver.a("step_factors").copy())
    np.save(os.path.join(out_dir, f"rms_{rank}.npy"), np.array(rms))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_equal_the_duplicated_mesh(tmp_path, oracle):
    world, sweeps = 2, 3
    mp.spawn(_worker, args=(world, _free_port(), sweeps, str(tmp_path)), nprocs=world, join=True)

    # single process: the reference's own -m 2 structure, each copy with its rank's initial state
    import ctypes as C
    lib = oracle.load()
    oc = oracle.OracleCase.from_input_dat(GOLDEN_INPUT, 2)
    L = oc.levels[0]
    lib.ora_adjust_ewt(L.coords, L.n_edges, L.edges)
    lib.ora_dampen_ewt(L.n_edges, L.edges, 5e-8)
    ff = oracle.farfield()
    nel1 = L.nel // 2
    q = oc.array(0, "variables").reshape(-1, 5)
    for r in range(world):
        rng = np.random.default_rng(500 + r)
        q[r * nel1:(r + 1) * nel1] = np.tile(np.array(ff.var), (nel1, 1)) * (1 + 0.02 * rng.uniform(-1, 1, (nel1, 5)))
    want_rms = []
    for _ in range(sweeps):
        oc.array(0, "old_variables")[:] = oc.array(0, "variables")
        lib.ora_compute_step_factor(L.nel, L.variables, L.volumes, L.step_factors)
        for j in range(3):
            lib.ora_compute_flux_edge(L.internal_start, L.n_internal, L.edges, L.variables, L.fluxes)
            lib.ora_compute_boundary_flux_edge(L.boundary_start, L.n_boundary, L.edges, L.variables, L.fluxes)
            lib.ora_compute_wall_flux_edge(L.wall_start, L.n_wall, L.edges, L.variables, L.fluxes, C.byref(ff))
            lib.ora_time_step(j, L.nel, L.step_factors, L.fluxes, L.old_variables, L.variables)
        lib.ora_residual(L.nel, L.old_variables, L.variables, L.residuals)
        want_rms.append(lib.ora_calc_rms(L.nel, L.residuals))
    want = oc.array(0, "variables").reshape(-1, 5)
    sf = oc.array(0, "step_factors")
    for r in range(world):
        got = np.load(tmp_path / f"vars_{r}.npy").reshape(-1, 5)
        # the global minimum makes both ranks use the same dt as the duplicated run; the numpy
        # stand-in for the first half of compute_step_factor may differ from libm's cbrt in the
        # last bit, hence 1e-15 rather than bitwise
        assert np.allclose(got, want[r * nel1:(r + 1) * nel1], rtol=1e-15, atol=0)
        assert np.allclose(np.load(tmp_path / f"sf_{r}.npy"), sf[r * nel1:(r + 1) * nel1], rtol=1e-15, atol=0)
        assert np.allclose(np.load(tmp_path / f"rms_{r}.npy"), want_rms, rtol=1e-12, atol=0)
    # and the coupling is real: without the all-reduce the two copies would use different dt
    sf0, sf1 = np.load(tmp_path / "sf_0.npy"), np.load(tmp_path / "sf_1.npy")
    assert np.array_equal(sf0, sf1)
