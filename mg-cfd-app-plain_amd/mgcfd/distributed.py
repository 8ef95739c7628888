"""Multi-GPU host logic: one process per GPU, one mesh copy (or mesh part) per rank.

The reference has no distributed path; the sharding below is the one its own `-m` mesh
duplication defines (src/Base/io_enhanced.cpp:89-201): the copies are independent except for
the GLOBAL minimum time step (src/Kernels/cfd_loops.cpp:137-150) and the RMS that is summed over
all nodes (src/Kernels/validation.cpp:91-105).  So a sweep needs exactly one all-reduce(MIN) of
one fp64 and, when the RMS is wanted, one all-reduce(SUM) of one fp64 — issued through
torch.distributed (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests) on
tensors that alias the solver's device scalars, between the two halves of compute_step_factor.
No other data moves between ranks.

The solver object only has to provide the kernel-granular calls of include/mgcfd.h plus two
scalar views; the product passes a :class:`mgcfd.api.Solver`, the CPU tests a stand-in.
"""
from __future__ import annotations

import math
import os

RK = 3


class DevScalarView:
    """Zero-copy torch view of an fp64 device scalar owned by libmgcfd_hip.so."""

    def __init__(self, ptr: int):
        self.__cuda_array_interface__ = {"shape": (1,), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}

    def tensor(self, device):
        import torch
        return torch.as_tensor(self, device=device)


class DevArrayView:
    """Zero-copy torch view of an fp64 device array owned by libmgcfd_hip.so."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}

    def tensor(self, device):
        import torch
        return torch.as_tensor(self, device=device)


class HipSolverAdapter:
    """Adds the two scalar views to a :class:`mgcfd.api.Solver` (device tensors aliasing library memory)."""

    def __init__(self, solver, device):
        self.s = solver
        self.device = device
        self._min = {}

    @property
    def stream_handle(self):
        return getattr(self.s, "_stream_handle", None)

    def __getattr__(self, name):
        return getattr(self.s, name)

    def min_tensor(self, level):
        if level not in self._min:
            self._min[level] = DevScalarView(self.s.step_factor_min_devptr(level)).tensor(self.device)
        return self._min[level]

    def partial_min_tensor(self, level):
        """The level's per-workgroup step-factor minima (a few KB): all-reducing THEM (MIN, element-wise) instead of
        the scalar saves the reduction kernel in front of the collective."""
        if ("p", level) not in self._min:
            ptr, count = self.s.step_factor_partials_devptr(level)
            self._min[("p", level)] = DevArrayView(ptr, count).tensor(self.device)
        return self._min[("p", level)]

    def sumsq_tensor(self, level):
        # launches the reduction; the returned tensor aliases its device result
        return DevScalarView(self.s.residual_sumsq_devptr(level)).tensor(self.device)

    def accept_restricted(self, fine, tensor):
        self.s.accept_restricted(fine, tensor.data_ptr())

    def array_tensor(self, level, name):
        """A whole node array of a level as the library holds it (the address is asked for every time: the state
        buffers rotate with every sweep)."""
        ptr, count = self.s.array_devptr(level, name)
        return DevArrayView(ptr, count).tensor(self.device)


def _check_current_stream(solver):
    """Kernels and collectives are ordered only if torch's CURRENT stream is the one the solver launches on
    (mgcfd_set_stream); a HIP solver that knows its stream is checked, stand-ins without one are not."""
    h = getattr(solver, "stream_handle", None)
    if h is None:
        return
    import torch
    if torch.cuda.is_available() and int(torch.cuda.current_stream().cuda_stream) != int(h):
        raise RuntimeError("torch's current stream is not the solver's stream: call torch.cuda.set_stream(st) and solver.set_stream(st.cuda_stream) first")


class ShardedSweep:
    """Smoothing sweeps on per-rank mesh copies coupled through the global-min time step."""

    def __init__(self, solver, dist=None, global_time_step: bool = True, fused: bool = True):
        self.solver = solver
        self.fused = fused and hasattr(solver, "sweep_begin")   # one launch per RK stage (mgcfd_sweep_begin/_end)
        self.overlap_even_alone = False     # tests: take the sweep_flux0 path without a process group
        self.reduce_partials = True         # all-reduce the partial minima (see sweep)
        # all-reduce a torch-owned copy instead of the library's memory (fallback, see sweep; MGCFD_ALLREDUCE_STAGED=1 forces it)
        self._staged = os.environ.get("MGCFD_ALLREDUCE_STAGED") == "1"
        self.dist = dist if (dist is not None and dist.is_initialized() and dist.get_world_size() > 1) else None
        self.global_time_step = global_time_step      # False for mesh_name = fvcorr (local time step)
        if self.dist:
            _check_current_stream(solver)

    @property
    def world(self) -> int:
        return self.dist.get_world_size() if self.dist else 1

    def sweep(self, level: int = 0):
        """copy, compute_step_factor (global min over ALL ranks), RK x (fluxes, time_step), residual:
        the per-level body of the reference's cycle loop, src/euler3d_cpu_double.cpp:383-508."""
        s = self.solver
        if self.fused:
            # fused kernels: everything before the collective, the collective, everything after.  Where the solver
            # offers them, the per-workgroup PARTIAL minima (a few KB) are all-reduced instead of the scalar: no
            # reduction kernel in front of the collective, the first stage takes the minimum over the global partials.
            # (up to ~2,000 tiles: every workgroup of the first stage reduces the whole array, which is quadratic)
            partials = self.reduce_partials and hasattr(s, "partial_min_tensor") and s.partial_min_tensor(level).numel() <= 2048
            (s.sweep_begin_partials if partials else s.sweep_begin)(level)
            if self.global_time_step and self.dist:
                # The first stage's fluxes do not depend on the time step: run them while the all-reduce
                # (latency bound, tens of microseconds over xGMI) is in flight.
                t = s.partial_min_tensor(level) if partials else s.min_tensor(level)
                # in place on the library's own device memory (the tensor aliases it), or — MGCFD_ALLREDUCE_STAGED=1, chosen
                # up front and therefore by every rank alike — through a torch-owned copy (two small device copies per sweep,
                # same values) for a backend that refuses memory torch did not allocate.  A failure of the collective itself is
                # NOT retried here: one rank re-issuing a collective the others never issue leaves the ranks out of step for
                # good, and after an asynchronous RCCL error the communicator is unusable anyway — the error propagates.
                stage = t.clone() if self._staged else None
                work = self.dist.all_reduce(stage if self._staged else t, op=self.dist.ReduceOp.MIN, async_op=True)
                s.sweep_flux0(level)
                work.wait()          # stream-level wait: later kernels are ordered after the collective
                if stage is not None:
                    t.copy_(stage)
            elif self.overlap_even_alone:
                s.sweep_flux0(level)
            (s.sweep_end_partials if partials else s.sweep_end)(level)
            return
        s.copy_old_variables(level)
        if self.global_time_step:
            s.step_factor_local(level)
            if self.dist:
                self.dist.all_reduce(s.min_tensor(level), op=self.dist.ReduceOp.MIN)
            s.step_factor_apply(level)
        else:
            s.compute_step_factor(level)
        for j in range(RK):
            s.compute_fluxes(level)
            s.time_step(level, j)
        s.residual(level)

    def rms(self, level: int, nel_local: int) -> float:
        """calc_rms over every rank's nodes: sqrt(sum of squares / total nel)."""
        t = self.solver.sumsq_tensor(level)
        if self.dist:
            t = t.clone()
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return math.sqrt(float(t.item()) / (nel_local * self.world))


class PartitionedSweep:
    """Smoothing sweeps on ONE level partitioned over ranks (mgcfd.partition): besides the
    global-min time step, every Runge-Kutta stage ends with a halo exchange of `variables`
    (40 B per halo node and neighbouring pair) — point-to-point sends over xGMI through RCCL
    (torch.distributed.batch_isend_irecv), packed/unpacked on the GPU by mgcfd_halo_pack/_unpack.

    `exchange` moves the messages: the default uses torch.distributed; tests inject an in-process
    copier to run several parts on one GPU."""

    def __init__(self, solver, part, dist=None, make_buffer=None, exchange=None, allreduce_min=None,
                 global_time_step=True, fused=False):
        self.s = solver
        self.fused = fused and hasattr(solver, "sweep_stage")     # one fused launch per RK stage (mgcfd_sweep_stage)
        self.allreduce_min_fn = allreduce_min
        self.part = part
        self.dist = dist if (dist is not None and dist.is_initialized() and dist.get_world_size() > 1) else None
        if self.dist:
            _check_current_stream(solver)
        self.global_time_step = global_time_step
        self.exchange_fn = exchange or self._exchange_torch
        self.peers = sorted(set(part.send) | set(part.recv))
        self.plan_send = {p: solver.halo_plan(0, part.send[p]) for p in part.send}
        self.plan_recv = {p: solver.halo_plan(0, part.recv[p]) for p in part.recv}
        self.buf_send = {p: make_buffer(len(part.send[p]) * 5) for p in part.send}
        self.buf_recv = {p: make_buffer(len(part.recv[p]) * 5) for p in part.recv}

    def _exchange_torch(self, sweep):
        d = self.dist
        if d is None:
            return
        staged = any(_gloo_on_device(d, b) for b in list(self.buf_send.values()) + list(self.buf_recv.values()))
        if staged:
            import torch
            torch.cuda.current_stream().synchronize()          # the pack kernels have written the messages
            send = {p: b.cpu() for p, b in self.buf_send.items()}
            recv = {p: torch.empty(b.shape, dtype=b.dtype) for p, b in self.buf_recv.items()}
        else:
            send, recv = self.buf_send, self.buf_recv
        ops = []
        for p in self.peers:
            if p in send:
                ops.append(d.P2POp(d.isend, send[p], p))
            if p in recv:
                ops.append(d.P2POp(d.irecv, recv[p], p))
        for req in d.batch_isend_irecv(ops):
            req.wait()
        if staged:
            for p, h in recv.items():
                self.buf_recv[p].copy_(h)
            torch.cuda.current_stream().synchronize()

    def exchange(self, name="variables"):
        s = self.s
        for p, plan in self.plan_send.items():
            s.halo_pack(0, plan, name, self.buf_send[p].data_ptr())
        self.exchange_fn(self)
        for p, plan in self.plan_recv.items():
            s.halo_unpack(0, plan, name, self.buf_recv[p].data_ptr())

    def sweep(self):
        """The per-level body of the reference's cycle loop (src/euler3d_cpu_double.cpp:383-508) on a
        partitioned level; ghosts must be current on entry (call exchange() after setting the state)."""
        s = self.s
        if self.fused:
            _fused_partitioned_sweep(s, 0, self.global_time_step, self.allreduce_min_fn, self.dist,
                                     lambda: self.exchange("stage"), self)
            return
        s.copy_old_variables(0)
        if self.global_time_step:
            s.step_factor_local(0)
            if self.allreduce_min_fn:
                self.allreduce_min_fn(self)
            elif self.dist:
                _all_reduce_min(self.dist, s.min_tensor(0))
            s.step_factor_apply(0)
        else:
            s.compute_step_factor(0)
        for j in range(RK):
            s.compute_fluxes(0)
            s.time_step(0, j)
            self.exchange("variables")
        s.residual(0)


class LevelPerRankCycle:
    """One multigrid level per rank (BASELINE config 4): rank r owns level l when rank_of(l) == r.  The V-cycle is
    sequential in levels, so this buys placement, not concurrency; what moves is what the reference's cycle loop hands
    from level to level (src/euler3d_cpu_double.cpp:527-688): the restricted coarse `variables` on the way up (computed
    where the fine level lives, 40 B per coarse node) and the coarse `residuals` on the way down — whole-array
    point-to-point messages between two solvers built from the same level data (torch.distributed send/recv: RCCL over
    one xGMI link on the GPUs, gloo in the CPU tests).

    `solver` holds every level (only the owned ones are ever swept) and provides smooth / restrict / prolong plus
    array_tensor(level, name) -> tensor aliasing the level's array and array_written(level, name)."""

    def __init__(self, solver, n_levels, rank, world, rank_of=None, send=None, recv=None, dist=None):
        self.s, self.n, self.rank, self.world = solver, n_levels, rank, world
        self.rank_of = rank_of or (lambda l: l % world)
        self.dist = dist
        def _send(t, dst):
            if _gloo_on_device(dist, t):                    # (see _gloo_on_device: staged through the host, stream drained)
                import torch
                torch.cuda.current_stream().synchronize()
                dist.send(t.cpu(), dst)
            else:
                dist.send(t, dst)

        def _recv(t, src):
            if _gloo_on_device(dist, t):
                import torch
                torch.cuda.current_stream().synchronize()
                h = torch.empty(t.shape, dtype=t.dtype)
                dist.recv(h, src)
                t.copy_(h)
                torch.cuda.current_stream().synchronize()
            else:
                dist.recv(t, src)
        self.send = send or _send
        self.recv = recv or _recv
        self._stage = {}

    def _hand_over(self, level, name, src, dst):
        if src == dst:
            return
        if self.rank == src:
            self.send(self.s.array_tensor(level, name), dst)
        elif self.rank == dst:
            if name == "variables":
                # restricted variables: mg_restrict leaves a coarse node without children at ITS old value
                # (src/Kernels/mg_loops.cpp:63-78,174-189) and only this rank, which sweeps the coarse level, has it:
                # receive into a staging array and take the nodes that have children
                t = self.s.array_tensor(level, name)
                if level not in self._stage:
                    self._stage[level] = t.clone()
                self.recv(self._stage[level], src)
                self.s.accept_restricted(level - 1, self._stage[level])
            else:
                self.recv(self.s.array_tensor(level, name), src)
            self.s.array_written(level, name)

    def cycle(self):
        """One multigrid cycle: sweeps on levels 0..n-1, n-2..1, restrictions going up, prolongations coming down."""
        s, n, me = self.s, self.n, self.rank
        for l in range(n):
            owner = self.rank_of(l)
            if me == owner:
                s.smooth(l, 1)
            if l + 1 < n:
                if me == owner:
                    s.restrict(l)                                   # fills level l+1's variables HERE
                self._hand_over(l + 1, "variables", owner, self.rank_of(l + 1))
        for l in range(n - 2, -1, -1):
            owner = self.rank_of(l)
            self._hand_over(l + 1, "residuals", self.rank_of(l + 1), owner)
            if me == owner:
                s.prolong(l)
                if l > 0:
                    s.smooth(l, 1)


def _gloo_on_device(dist, t) -> bool:
    """gloo given a DEVICE tensor (bench.py's one-GPU rehearsal): its point-to-point calls know nothing of streams — a send
    may read the buffer before the pack kernel has written it (tools/torch_path_check.py: the sweeps then differ from the
    whole level in most runs) — so such messages are staged through the host with the stream drained on both sides."""
    try:
        return bool(getattr(t, "is_cuda", False)) and dist.get_backend() == "gloo"
    except Exception:
        return False


def _all_reduce_min(dist, t):
    if _gloo_on_device(dist, t):
        import torch
        torch.cuda.current_stream().synchronize()
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MIN)
        t.copy_(h)
        torch.cuda.current_stream().synchronize()
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MIN)


def _fused_partitioned_sweep(s, level, global_time_step, allreduce_min_fn, dist, exchange_stage, owner):
    """A sweep of a partitioned level with one fused launch per Runge-Kutta stage: the first half of
    compute_step_factor reduced to the rank's scalar minimum (the ranks' levels differ in size, so their per-workgroup
    minima cannot be all-reduced element-wise), the all-reduce(MIN) of that scalar, then stage / halo message three
    times (the message carries the state the stage just wrote, MGCFD_ARR_STAGE)."""
    s.sweep_begin(level)
    if global_time_step:
        if allreduce_min_fn:
            allreduce_min_fn(owner, level)
        elif dist:
            _all_reduce_min(dist, s.min_tensor(level))
    for j in range(RK):
        s.sweep_stage(level, j, False)
        exchange_stage()


class PartitionedCycle:
    """Multigrid V-cycles on a hierarchy partitioned over ranks (mgcfd.partition.partition_hierarchy): every level is
    split, each rank sweeps its owned nodes, and ghost values move wherever the next operation reads them —
    `variables` after every time_step, after mgcfd_restrict (the coarse level's ghosts) and after mgcfd_prolong (the
    fine level's), coarse `residuals` before mgcfd_prolong — as packed point-to-point messages (mgcfd_halo_pack /
    _unpack around torch.distributed.batch_isend_irecv, or the injected `exchange`).  One all-reduce(MIN) of the time
    step per sweep as in PartitionedSweep."""

    def __init__(self, solver, hpart, dist=None, make_buffer=None, exchange=None, allreduce_min=None, fused=False):
        self.s, self.h = solver, hpart
        self.fused = fused and hasattr(solver, "sweep_stage")
        self.dist = dist if (dist is not None and dist.is_initialized() and dist.get_world_size() > 1) else None
        self.exchange_fn = exchange
        self.allreduce_min_fn = allreduce_min
        self.n = len(hpart.levels)
        self.plan_send, self.plan_recv, self.buf_send, self.buf_recv = [], [], [], []
        for l, P in enumerate(hpart.levels):
            self.plan_send.append({p: solver.halo_plan(l, P.send[p]) for p in P.send})
            self.plan_recv.append({p: solver.halo_plan(l, P.recv[p]) for p in P.recv})
            self.buf_send.append({p: make_buffer(len(P.send[p]) * 5) for p in P.send})
            self.buf_recv.append({p: make_buffer(len(P.recv[p]) * 5) for p in P.recv})

    def exchange(self, level, name):
        s = self.s
        for p, plan in self.plan_send[level].items():
            s.halo_pack(level, plan, name, self.buf_send[level][p].data_ptr())
        if self.exchange_fn:
            self.exchange_fn(self, level)
        elif self.dist:
            d, ops = self.dist, []
            send, recv = self.buf_send[level], self.buf_recv[level]
            # (gloo given DEVICE tensors — bench.py's one-GPU rehearsal — knows nothing of streams: through the host, stream drained)
            staged = any(_gloo_on_device(d, b) for b in list(send.values()) + list(recv.values()))
            if staged:
                import torch
                torch.cuda.current_stream().synchronize()
                send = {p: b.cpu() for p, b in send.items()}
                recv = {p: torch.empty(b.shape, dtype=b.dtype) for p, b in recv.items()}
            for p in sorted(set(send) | set(recv)):
                if p in send:
                    ops.append(d.P2POp(d.isend, send[p], p))
                if p in recv:
                    ops.append(d.P2POp(d.irecv, recv[p], p))
            for req in d.batch_isend_irecv(ops):
                req.wait()
            if staged:
                for p, h in recv.items():
                    self.buf_recv[level][p].copy_(h)
                torch.cuda.current_stream().synchronize()
        for p, plan in self.plan_recv[level].items():
            s.halo_unpack(level, plan, name, self.buf_recv[level][p].data_ptr())

    def sweep(self, level):
        s = self.s
        if self.fused:
            _fused_partitioned_sweep(s, level, True, self.allreduce_min_fn, self.dist,
                                     lambda: self.exchange(level, "stage"), self)
            return
        s.copy_old_variables(level)
        s.step_factor_local(level)
        if self.allreduce_min_fn:
            self.allreduce_min_fn(self, level)
        elif self.dist:
            self.dist.all_reduce(s.min_tensor(level), op=self.dist.ReduceOp.MIN)
        s.step_factor_apply(level)
        for j in range(RK):
            s.compute_fluxes(level)
            s.time_step(level, j)
            self.exchange(level, "variables")
        s.residual(level)

    def cycle(self):
        """Sweeps on levels 0..n-1, n-2..1 with the transfers between them (src/euler3d_cpu_double.cpp:371-694)."""
        s, n = self.s, self.n
        for l in range(n):
            self.sweep(l)
            if l + 1 < n:
                s.restrict(l)
                self.exchange(l + 1, "variables")
        for l in range(n - 2, -1, -1):
            self.exchange(l + 1, "residuals")
            s.prolong(l)
            self.exchange(l, "variables")
            if l > 0:
                self.sweep(l)
