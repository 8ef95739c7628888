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


# ------------------------------------------------------------------------------------------
# Partitioned level + halo exchange over gloo point-to-point (BASELINE config 5 in miniature)
# ------------------------------------------------------------------------------------------
def _libm_cbrt(x):
    """cbrt through the C library the oracle itself calls (numpy's may differ in the last bit)."""
    import ctypes
    import ctypes.util
    libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    libm.cbrt.restype = ctypes.c_double
    libm.cbrt.argtypes = [ctypes.c_double]
    return np.array([libm.cbrt(float(v)) for v in np.asarray(x).ravel()]).reshape(np.shape(x))


class OraclePartSolver:
    """CPU stand-in (oracle-backed) for a partitioned mgcfd.api.Solver: one level, ghosts last."""

    def __init__(self, oracle, level, q):
        import ctypes as C
        self.O, self.C = oracle, C
        self.lib = oracle.load()
        self.nel = level["nel"]
        self.edges = np.ascontiguousarray(level["edges"]).copy()
        self.coords = np.ascontiguousarray(level["coords"], dtype=np.float64)
        self.lib.ora_adjust_ewt(oracle.ptr(self.coords), len(self.edges), oracle.ptr(self.edges))
        self.lib.ora_dampen_ewt(len(self.edges), oracle.ptr(self.edges), 5e-8)
        self.ni, self.nb, self.nw = level["n_internal"], level["n_boundary"], level["n_wall"]
        self.vol = np.ascontiguousarray(level["volumes"], dtype=np.float64)
        self.ff = oracle.farfield()
        self.v = np.ascontiguousarray(q, dtype=np.float64).copy()
        self.old, self.f, self.res = np.zeros_like(self.v), np.zeros_like(self.v), np.zeros_like(self.v)
        self.sf = np.zeros(self.nel)
        self._min = torch.zeros(1, dtype=torch.float64)
        self.plans = []

    def copy_old_variables(self, l):
        self.old[:] = self.v

    def step_factor_local(self, l):
        q = self.v
        vel = q[:, 1:4] / q[:, :1]
        sp2 = (vel[:, 0] * vel[:, 0] + vel[:, 1] * vel[:, 1]) + vel[:, 2] * vel[:, 2]
        p = (1.4 - 1.0) * (q[:, 4] - 0.5 * q[:, 0] * sp2)
        if not hasattr(self, "_cbrt_vol"):
            self._cbrt_vol = _libm_cbrt(self.vol)
        self._min[0] = float((0.5 * (self._cbrt_vol / (np.sqrt(sp2) + np.sqrt(1.4 * p / q[:, 0])))).min())

    def min_tensor(self, l):
        return self._min

    def step_factor_apply(self, l):
        self.sf[:] = float(self._min[0]) / self.vol

    def compute_fluxes(self, l):
        O, lib = self.O, self.lib
        lib.ora_compute_flux_edge(0, self.ni, O.ptr(self.edges), O.ptr(self.v), O.ptr(self.f))
        lib.ora_compute_boundary_flux_edge(self.ni, self.nb, O.ptr(self.edges), O.ptr(self.v), O.ptr(self.f))
        lib.ora_compute_wall_flux_edge(self.ni + self.nb, self.nw, O.ptr(self.edges), O.ptr(self.v), O.ptr(self.f), self.C.byref(self.ff))

    def time_step(self, l, j):
        O = self.O
        self.lib.ora_time_step(j, self.nel, O.ptr(self.sf), O.ptr(self.f), O.ptr(self.old), O.ptr(self.v))

    def residual(self, l):
        self.res[:] = self.v - self.old

    def halo_plan(self, l, ids):
        self.plans.append(np.asarray(ids, dtype=np.int64))
        return len(self.plans) - 1

    @staticmethod
    def _view(ptr, n):
        import ctypes
        return np.ctypeslib.as_array((ctypes.c_double * n).from_address(ptr))

    def halo_pack(self, l, plan, name, ptr):
        ids = self.plans[plan]
        if len(ids):
            self._view(ptr, len(ids) * 5)[:] = self.v[ids].ravel()

    def halo_unpack(self, l, plan, name, ptr):
        ids = self.plans[plan]
        if len(ids):
            self.v[ids] = self._view(ptr, len(ids) * 5).reshape(-1, 5)


def _part_worker(rank, world, port, sweeps, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    import mgcfd
    from mgcfd import meshgen
    from mgcfd.distributed import PartitionedSweep
    from mgcfd.partition import partition_level, slab_partition
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mg = meshgen.make_multigrid((10,), "m6wing", seed=4, cavity_radius=0.15, jitter=0.2, area_noise=0.05, volume_noise=0.05)
    L = mgcfd.generated_to_levels(mg)[0]
    P = partition_level(L, slab_partition(np.asarray(L["coords"]), world))[rank]
    ff = oracle_py.farfield()
    rng = np.random.default_rng(9)
    q0 = np.tile(np.array(ff.var), (L["nel"], 1)) * (1 + 0.02 * rng.uniform(-1, 1, (L["nel"], 5)))
    solver = OraclePartSolver(oracle_py, P.level, q0[P.global_ids])
    sw = PartitionedSweep(solver, P, dist, make_buffer=lambda n: torch.zeros(n, dtype=torch.float64))
    for _ in range(sweeps):
        sw.sweep()
    np.save(os.path.join(out_dir, f"part_vars_{rank}.npy"), solver.v[:P.n_owned])
    np.save(os.path.join(out_dir, f"part_ids_{rank}.npy"), P.global_ids[:P.n_owned])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_partitioned_level_over_gloo_equals_whole_mesh(tmp_path, oracle):
    import ctypes as C
    import mgcfd
    from mgcfd import meshgen
    world, sweeps = 2, 2
    mp.spawn(_part_worker, args=(world, _free_port(), sweeps, str(tmp_path)), nprocs=world, join=True)
    mg = meshgen.make_multigrid((10,), "m6wing", seed=4, cavity_radius=0.15, jitter=0.2, area_noise=0.05, volume_noise=0.05)
    L = mgcfd.generated_to_levels(mg)[0]
    ff = oracle.farfield()
    rng = np.random.default_rng(9)
    q0 = np.tile(np.array(ff.var), (L["nel"], 1)) * (1 + 0.02 * rng.uniform(-1, 1, (L["nel"], 5)))
    whole = OraclePartSolver(oracle, L, q0)
    for _ in range(sweeps):
        whole.copy_old_variables(0)
        whole.step_factor_local(0)
        whole.step_factor_apply(0)
        for j in range(3):
            whole.compute_fluxes(0)
            whole.time_step(0, j)
        whole.residual(0)
    seen = 0
    for r in range(world):
        ids = np.load(tmp_path / f"part_ids_{r}.npy")
        got = np.load(tmp_path / f"part_vars_{r}.npy")
        assert np.array_equal(got.view(np.int64), whole.v[ids].view(np.int64))     # same order of sums => same bits
        seen += len(ids)
    assert seen == L["nel"]


# ------------------------------------------------------------------------------------------
# One multigrid level per rank (BASELINE config 4): restricted variables up, residuals down
# ------------------------------------------------------------------------------------------
GOLDEN_3LVL = os.path.join(ROOT, "tests", "golden", "m6_3lvl")


class OracleMGSolver:
    """Oracle-backed stand-in for a multi-level mgcfd.api.Solver: smooth / restrict / prolong on the
    reference's own arrays, torch tensors aliasing them."""

    def __init__(self, oracle, input_dat=None):
        import ctypes as C
        self.O, self.C = oracle, C
        self.lib = oracle.load()
        self.case = oracle.OracleCase.from_input_dat(input_dat or os.path.join(GOLDEN_3LVL, "input", "input.dat"))
        self.ff = oracle.farfield()
        for l in range(self.case.nlevels):
            L = self.case.levels[l]
            self.lib.ora_adjust_ewt(L.coords, L.n_edges, L.edges)
            self.lib.ora_dampen_ewt(L.n_edges, L.edges, 5e-8)
            self.lib.ora_initialize_variables(L.nel, L.variables, C.byref(self.ff))
        self.written = []

    def smooth(self, l, sweeps=1):
        L, lib = self.case.levels[l], self.lib
        for _ in range(sweeps):
            self.case.array(l, "old_variables")[:] = self.case.array(l, "variables")
            lib.ora_compute_step_factor(L.nel, L.variables, L.volumes, L.step_factors)
            for j in range(3):
                lib.ora_compute_flux_edge(L.internal_start, L.n_internal, L.edges, L.variables, L.fluxes)
                lib.ora_compute_boundary_flux_edge(L.boundary_start, L.n_boundary, L.edges, L.variables, L.fluxes)
                lib.ora_compute_wall_flux_edge(L.wall_start, L.n_wall, L.edges, L.variables, L.fluxes, self.C.byref(self.ff))
                lib.ora_time_step(j, L.nel, L.step_factors, L.fluxes, L.old_variables, L.variables)
            lib.ora_residual(L.nel, L.old_variables, L.variables, L.residuals)

    def restrict(self, fine):
        F, Cc = self.case.levels[fine], self.case.levels[fine + 1]
        scratch = np.zeros(Cc.nel, dtype=np.int64)
        self.lib.ora_mg_restrict(F.variables, Cc.variables, Cc.nel, F.mg_map, self.O.ptr(scratch), F.mgc)

    def prolong(self, fine):
        F, Cc = self.case.levels[fine], self.case.levels[fine + 1]
        self.lib.ora_prolong_residuals_interpolate_proper(F.edges, F.n_internal, Cc.residuals, F.residuals, F.variables,
                                                           F.nel, F.mg_map, Cc.coords, F.coords)

    def array_tensor(self, l, name):
        return torch.from_numpy(self.case.array(l, name))

    def accept_restricted(self, fine, tensor):
        """mgcfd_accept_restricted: coarse nodes with children take the message, the others keep their value."""
        m = np.asarray(self.case.mg_map(fine))
        has = np.zeros(self.case.levels[fine + 1].nel, dtype=bool)
        has[m] = True
        v = self.case.array(fine + 1, "variables").reshape(-1, 5)
        v[has] = tensor.numpy().reshape(-1, 5)[has]

    def array_written(self, l, name):
        self.written.append((l, name))


def _level_worker(rank, world, port, cycles, out_dir, input_dat=None):
    sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    from mgcfd.distributed import LevelPerRankCycle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    solver = OracleMGSolver(oracle_py, input_dat)
    cyc = LevelPerRankCycle(solver, solver.case.nlevels, rank, world, dist=dist)
    for _ in range(cycles):
        cyc.cycle()
    for l in range(solver.case.nlevels):
        if cyc.rank_of(l) == rank:
            np.save(os.path.join(out_dir, f"lvl{l}.npy"), solver.case.array(l, "variables").copy())
    np.save(os.path.join(out_dir, f"written_{rank}.npy"), np.array([f"{l}:{n}" for l, n in solver.written]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_one_level_per_rank_equals_the_reference_binary(tmp_path, oracle):
    """Three levels on two ranks (levels 0 and 2 on rank 0, level 1 on rank 1): after the golden case's two
    cycles level 0 must equal the reference binary's dump bit for bit, and every level the single-process oracle."""
    world, cycles = 2, 2
    mp.spawn(_level_worker, args=(world, _free_port(), cycles, str(tmp_path)), nprocs=world, join=True)
    want = np.loadtxt(os.path.join(GOLDEN_3LVL, "variables.level0.txt"))
    got = np.load(tmp_path / "lvl0.npy").reshape(-1, 5)
    assert np.array_equal(got.view(np.int64), want.view(np.int64))
    oc = oracle.OracleCase.from_input_dat(os.path.join(GOLDEN_3LVL, "input", "input.dat"))
    rc, _, _ = oc.solve(cycles, run_indirect_rw=False)
    assert rc == 0
    for l in range(oc.nlevels):
        assert np.array_equal(np.load(tmp_path / f"lvl{l}.npy").view(np.int64), oc.array(l, "variables").view(np.int64)), l
    # the receiving side told its solver about every array it received
    assert set(np.load(tmp_path / "written_1.npy")) == {"1:variables", "2:residuals"}
    assert set(np.load(tmp_path / "written_0.npy")) == {"2:variables", "1:residuals"}


@pytest.mark.timeout(300)
def test_one_level_per_rank_with_childless_coarse_nodes(tmp_path, oracle):
    """A hierarchy whose middle level has coarse nodes WITHOUT children (10^3 -> 9^3 nearest-node map: 11 of them):
    mg_restrict leaves such a node at its old value, which only the rank that sweeps the coarse level holds — the
    restricted array that arrives from the finer level's rank must not overwrite it (mgcfd_accept_restricted).  Three
    levels on three ranks, three cycles, every level against the single-process oracle."""
    from mgcfd import meshgen
    world, cycles = 3, 3
    d = tmp_path / "mesh"
    d.mkdir()
    mg = meshgen.make_multigrid((10, 9, 5), "m6wing", seed=3, jitter=0.2, area_noise=0.05, volume_noise=0.05)
    meshgen.write_input(mg, str(d))
    dat = str(d / "input.dat")
    oc = oracle.OracleCase.from_input_dat(dat)
    m = np.asarray(oc.mg_map(0))
    assert len(np.unique(m)) < oc.levels[1].nel, "the case must have childless coarse nodes"
    out = tmp_path / "out"
    out.mkdir()
    mp.spawn(_level_worker, args=(world, _free_port(), cycles, str(out), dat), nprocs=world, join=True)
    rc, _, _ = oc.solve(cycles, run_indirect_rw=False)
    assert rc == 0
    for l in range(oc.nlevels):
        assert np.array_equal(np.load(out / f"lvl{l}.npy").view(np.int64), oc.array(l, "variables").view(np.int64)), l


# ------------------------------------------------------------------------------------------
# A whole hierarchy partitioned over two ranks: multigrid transfers across the partition
# ------------------------------------------------------------------------------------------
class OracleHierarchyPartSolver:
    """Oracle-backed stand-in for a solver created by mgcfd_create_partitioned_mg: per level the single-level
    stand-in above, plus restrict (children averaged in GLOBAL-id order, as order_keys asks) and prolong."""

    def __init__(self, oracle, hpart, ff_var):
        self.O = oracle
        self.lib = oracle.load()
        self.h = hpart
        self.lv = []
        for P in hpart.levels:
            q = np.tile(np.array(ff_var), (P.level["nel"], 1))
            self.lv.append(OraclePartSolver(oracle, P.level, q))

    # ---- per-level loops: delegate ----
    def copy_old_variables(self, l): self.lv[l].copy_old_variables(0)
    def step_factor_local(self, l): self.lv[l].step_factor_local(0)
    def min_tensor(self, l): return self.lv[l].min_tensor(0)
    def step_factor_apply(self, l): self.lv[l].step_factor_apply(0)
    def compute_fluxes(self, l): self.lv[l].compute_fluxes(0)
    def time_step(self, l, j): self.lv[l].time_step(0, j)
    def residual(self, l): self.lv[l].residual(0)

    def halo_plan(self, l, ids): return self.lv[l].halo_plan(0, ids)

    def _field(self, l, name): return self.lv[l].v if name == "variables" else self.lv[l].res

    def halo_pack(self, l, plan, name, ptr):
        ids = self.lv[l].plans[plan]
        if len(ids):
            OraclePartSolver._view(ptr, len(ids) * 5)[:] = self._field(l, name)[ids].ravel()

    def halo_unpack(self, l, plan, name, ptr):
        ids = self.lv[l].plans[plan]
        if len(ids):
            self._field(l, name)[ids] = OraclePartSolver._view(ptr, len(ids) * 5).reshape(-1, 5)

    def restrict(self, fine):
        F, Cc = self.h.levels[fine], self.h.levels[fine + 1]
        m = F.level["mg_map"]
        qf, qc = self.lv[fine].v, self.lv[fine + 1].v
        order = np.argsort(F.global_ids, kind="stable")                  # children summed by global id
        acc = np.zeros_like(qc)
        cnt = np.zeros(len(qc), dtype=np.int64)
        for i in order:
            acc[m[i]] += qf[i]
            cnt[m[i]] += 1
        has = cnt > 0
        qc[has] = acc[has] * (1.0 / cnt[has])[:, None]

    def prolong(self, fine):
        O, lib = self.O, self.lib
        F, Cc = self.lv[fine], self.lv[fine + 1]
        m = np.ascontiguousarray(self.h.levels[fine].level["mg_map"], dtype=np.int64)
        n_owned = self.h.levels[fine].n_owned
        keep = F.v[n_owned:].copy()
        lib.ora_prolong_residuals_interpolate_proper(O.ptr(F.edges), F.ni, O.ptr(Cc.res), O.ptr(F.res), O.ptr(F.v),
                                                      F.nel, O.ptr(m), O.ptr(Cc.coords), O.ptr(F.coords))
        F.v[n_owned:] = keep                                             # ghosts: whatever, the exchange overwrites them


def _hier_worker(rank, world, port, cycles, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    import mgcfd
    from mgcfd import meshgen
    from mgcfd.distributed import PartitionedCycle
    from mgcfd.partition import partition_hierarchy, rcb_partition
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mg = meshgen.make_multigrid((8, 4), "m6wing", seed=4, jitter=0.2, area_noise=0.05, volume_noise=0.05)
    levels = mgcfd.generated_to_levels(mg)
    H = partition_hierarchy(levels, rcb_partition(np.asarray(levels[0]["coords"]), world))[rank]
    solver = OracleHierarchyPartSolver(oracle_py, H, oracle_py.farfield().var)
    cyc = PartitionedCycle(solver, H, dist, make_buffer=lambda n: torch.zeros(max(n, 1), dtype=torch.float64))
    for _ in range(cycles):
        cyc.cycle()
    for l, P in enumerate(H.levels):
        np.save(os.path.join(out_dir, f"h_vars_{rank}_{l}.npy"), solver.lv[l].v[:P.n_owned])
        np.save(os.path.join(out_dir, f"h_ids_{rank}_{l}.npy"), P.global_ids[:P.n_owned])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_partitioned_hierarchy_over_gloo_equals_whole_mesh(tmp_path, oracle):
    """Two levels split over two ranks, ghost values moved by real gloo point-to-point messages after every time_step,
    restrict and prolong: after two V-cycles every level's owned nodes equal the single-process oracle run bit for bit."""
    import mgcfd
    from mgcfd import meshgen
    world, cycles = 2, 2
    mp.spawn(_hier_worker, args=(world, _free_port(), cycles, str(tmp_path)), nprocs=world, join=True)
    mg = meshgen.make_multigrid((8, 4), "m6wing", seed=4, jitter=0.2, area_noise=0.05, volume_noise=0.05)
    levels = mgcfd.generated_to_levels(mg)
    # single process, the oracle's own driver on the whole hierarchy
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
    assert lib.ora_solve(lv, n, mg.mesh_variant, cycles, 0, oracle.ptr(rms), None) == 0
    want = [k[3][0] for k in keep if isinstance(k, tuple)]
    for l in range(n):
        seen = 0
        for r in range(world):
            ids = np.load(tmp_path / f"h_ids_{r}_{l}.npy")
            got = np.load(tmp_path / f"h_vars_{r}_{l}.npy")
            assert np.array_equal(got.view(np.int64), want[l][ids].view(np.int64)), (l, r)
            seen += len(ids)
        assert seen == levels[l]["nel"]
