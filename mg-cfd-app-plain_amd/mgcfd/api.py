"""ctypes binding of libmgcfd_hip.so (include/mgcfd.h).

The Python layer is plumbing for tests, bench.py and torch.distributed — the product is
the HIP library.  There is no CPU fallback: if the shared object is missing or no GPU is
present, construction of a :class:`Solver` raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import List, Optional, Sequence

import numpy as np

from .meshgen import EDGE_DTYPE, LevelMesh, MultigridMesh, to_edge_arrays

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.normpath(os.path.join(PKG_DIR, "..", "csrc"))
# MGCFD_LIB overrides the library path (profiling builds; tools/exp_variants.py)
LIB_PATH = os.environ.get("MGCFD_LIB") or os.path.join(CSRC_DIR, "libmgcfd_hip.so")

NVAR = 5
RK = 3
LOOPS = ("flux", "update", "compute_step", "time_step", "restrict", "prolong", "indirect_rw")
ARR = {"variables": 0, "old_variables": 1, "fluxes": 2, "residuals": 3, "step_factors": 4, "volumes": 5, "stage": 6}
OPT = {"exact": 0, "timing": 1, "indirect_rw": 2, "check_invalid": 3, "flux_variant": 4, "fuse_update": 5, "graph": 6, "rank_split": 7}
ERR_NAMES = {0: "OK", 1: "ERR_ARG", 2: "ERR_IO", 3: "ERR_HIP", 4: "ERR_NAN", 5: "ERR_NEG_DENSITY",
             6: "ERR_NEG_ENERGY", 7: "ERR_VALIDATION"}

_vp = C.c_void_p
_i64 = C.c_int64


class LevelDesc(C.Structure):
    _fields_ = [("nel", _i64), ("n_edges", _i64), ("n_internal", _i64), ("n_boundary", _i64),
                ("n_wall", _i64), ("internal_start", _i64), ("boundary_start", _i64), ("wall_start", _i64),
                ("volumes", _vp), ("coords", _vp), ("edges", _vp), ("mg_map", _vp), ("mgc", _i64)]


class MgcfdError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {message}")
        self.code = code


# Every symbol include/mgcfd.h declares: (name, restype, argtypes)
_SIGNATURES = [
    ("mgcfd_last_error", C.c_char_p, []),
    ("mgcfd_abi_version", C.c_int, []),
    ("mgcfd_mesh_load", C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(_vp)]),
    ("mgcfd_mesh_load_ex", C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(_vp)]),
    ("mgcfd_mesh_free", None, [_vp]),
    ("mgcfd_mesh_num_levels", C.c_int, [_vp]),
    ("mgcfd_mesh_variant", C.c_int, [_vp]),
    ("mgcfd_mesh_size", C.c_int, [_vp]),
    ("mgcfd_mesh_level", C.c_int, [_vp, C.c_int, C.POINTER(LevelDesc)]),
    ("mgcfd_write_array", C.c_int, [C.c_char_p, _vp, _i64, C.c_int]),
    ("mgcfd_identify_differences", C.c_int, [_vp, _vp, _i64, C.c_int, C.POINTER(_i64)]),
    ("mgcfd_create", C.c_int, [C.POINTER(LevelDesc), C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    ("mgcfd_plan_audit", C.c_int, [C.POINTER(LevelDesc), C.c_int, C.c_int, C.POINTER(_i64), C.POINTER(C.POINTER(_i64)), C.c_char_p, _i64]),
    ("mgcfd_create_partitioned_mg", C.c_int, [C.POINTER(LevelDesc), C.c_int, C.c_int, C.c_int, C.POINTER(_i64),
                                                 C.POINTER(C.POINTER(_i64)), C.POINTER(_vp)]),
    ("mgcfd_create_from_mesh", C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    ("mgcfd_create_partitioned", C.c_int, [C.POINTER(LevelDesc), C.c_int, C.c_int, C.c_int, C.POINTER(_i64), C.POINTER(_vp)]),
    ("mgcfd_halo_plan", C.c_int, [_vp, C.c_int, _i64, _vp, C.POINTER(C.c_int)]),
    ("mgcfd_halo_pack", C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp]),
    ("mgcfd_halo_unpack", C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp]),
    ("mgcfd_destroy", None, [_vp]),
    ("mgcfd_set_option", C.c_int, [_vp, C.c_int, C.c_int]),
    ("mgcfd_level_has_edge_once", C.c_int, [_vp, C.c_int, C.POINTER(C.c_int)]),
    ("mgcfd_level_has_half_rows", C.c_int, [_vp, C.c_int, C.POINTER(C.c_int)]),
    ("mgcfd_pending_invalid_state", C.c_int, [_vp, C.POINTER(C.c_int64)]),
    ("mgcfd_level_has_order_free", C.c_int, [_vp, C.c_int, C.POINTER(C.c_int)]),
    ("mgcfd_level_tiling", C.c_int, [_vp, C.c_int, C.POINTER(C.c_int64)]),
    ("mgcfd_invalid_state_location", C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    ("mgcfd_get_option", C.c_int, [_vp, C.c_int, C.POINTER(C.c_int)]),
    ("mgcfd_set_stream", C.c_int, [_vp, _vp]),
    ("mgcfd_synchronize", C.c_int, [_vp]),
    ("mgcfd_num_levels", C.c_int, [_vp]),
    ("mgcfd_level_nel", _i64, [_vp, C.c_int]),
    ("mgcfd_level_num_internal_edges", _i64, [_vp, C.c_int]),
    ("mgcfd_get_far_field", C.c_int, [_vp, _vp]),
    ("mgcfd_copy_old_variables", C.c_int, [_vp, C.c_int]),
    ("mgcfd_compute_step_factor", C.c_int, [_vp, C.c_int]),
    ("mgcfd_compute_flux_edge", C.c_int, [_vp, C.c_int]),
    ("mgcfd_compute_boundary_flux_edge", C.c_int, [_vp, C.c_int]),
    ("mgcfd_compute_wall_flux_edge", C.c_int, [_vp, C.c_int]),
    ("mgcfd_compute_fluxes", C.c_int, [_vp, C.c_int]),
    ("mgcfd_time_step", C.c_int, [_vp, C.c_int, C.c_int]),
    ("mgcfd_zero_fluxes", C.c_int, [_vp, C.c_int]),
    ("mgcfd_indirect_rw", C.c_int, [_vp, C.c_int]),
    ("mgcfd_residual", C.c_int, [_vp, C.c_int]),
    ("mgcfd_calc_rms", C.c_int, [_vp, C.c_int, C.POINTER(C.c_double)]),
    ("mgcfd_check_for_invalid_variables", C.c_int, [_vp, C.c_int, C.POINTER(_i64)]),
    ("mgcfd_restrict", C.c_int, [_vp, C.c_int]),
    ("mgcfd_prolong", C.c_int, [_vp, C.c_int]),
    ("mgcfd_smooth", C.c_int, [_vp, C.c_int, C.c_int]),
    ("mgcfd_run_cycles", C.c_int, [_vp, C.c_int, _vp]),
    ("mgcfd_get_array", C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    ("mgcfd_set_array", C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    ("mgcfd_array_devptr", C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(_vp), C.POINTER(C.c_int64)]),
    ("mgcfd_array_written", C.c_int, [_vp, C.c_int, C.c_int]),
    ("mgcfd_get_edges", C.c_int, [_vp, C.c_int, _vp]),
    ("mgcfd_accept_restricted", C.c_int, [_vp, C.c_int, _vp]),
    ("mgcfd_get_loop_iters", C.c_int, [_vp, C.c_int, _vp]),
    ("mgcfd_get_loop_times", C.c_int, [_vp, C.c_int, _vp]),
    ("mgcfd_reset_monitoring", C.c_int, [_vp]),
    ("mgcfd_get_flux_kernel_time", C.c_int, [_vp, C.c_int, C.POINTER(C.c_double), C.POINTER(_i64)]),
    ("mgcfd_bench_flux", C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    ("mgcfd_bench_indirect_rw", C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    ("mgcfd_bench_stream_ceiling", C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    ("mgcfd_device_warm_up", C.c_int, [C.c_int]),
    ("mgcfd_step_factor_local", C.c_int, [_vp, C.c_int]),
    ("mgcfd_step_factor_min_devptr", C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    ("mgcfd_step_factor_partials_devptr", C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(C.c_int)]),
    ("mgcfd_sweep_stage", C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    ("mgcfd_sweep_begin_partials", C.c_int, [_vp, C.c_int]),
    ("mgcfd_sweep_end_partials", C.c_int, [_vp, C.c_int]),
    ("mgcfd_step_factor_apply", C.c_int, [_vp, C.c_int]),
    ("mgcfd_sweep_begin", C.c_int, [_vp, C.c_int]),
    ("mgcfd_sweep_flux0", C.c_int, [_vp, C.c_int]),
    ("mgcfd_sweep_end", C.c_int, [_vp, C.c_int]),
    ("mgcfd_residual_sumsq", C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    ("mgcfd_rccl_unique_id", C.c_int, [_vp]),
    ("mgcfd_rank_attach_rccl", C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    ("mgcfd_rank_detach", C.c_int, [_vp]),
    ("mgcfd_rank_set_halo", C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(_i64), C.POINTER(_vp), C.POINTER(_i64), C.POINTER(_vp)]),
    ("mgcfd_rank_halo_info", C.c_int, [_vp, C.c_int, C.POINTER(_i64)]),
    ("mgcfd_rank_exchange", C.c_int, [_vp, C.c_int]),
    ("mgcfd_rank_sweeps", C.c_int, [_vp, C.c_int, C.c_int]),
    ("mgcfd_rank_residual_sumsq", C.c_int, [_vp, C.c_int, C.POINTER(C.c_double)]),
    ("mgcfd_group_create", C.c_int, [C.c_int, C.POINTER(_vp), C.POINTER(_vp)]),
    ("mgcfd_group_destroy", None, [_vp]),
    ("mgcfd_group_exchange", C.c_int, [_vp, C.c_int]),
    ("mgcfd_rank_attach_plain", C.c_int, [_vp, C.c_int, C.c_int]),
    ("mgcfd_rank_ipc_export_size", C.c_int, [_vp, C.c_int, C.POINTER(C.c_int64)]),
    ("mgcfd_rank_ipc_export", C.c_int, [_vp, C.c_int, _vp]),
    ("mgcfd_rank_ipc_attach", C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    ("mgcfd_rank_ipc_status", C.c_int, [_vp, C.c_int, C.POINTER(C.c_int)]),
    ("mgcfd_rank_info", C.c_int, [_vp, C.POINTER(C.c_int)]),
    ("mgcfd_rank_graph_status", C.c_int, [_vp, C.c_int, C.POINTER(_i64)]),
    ("mgcfd_group_cycles", C.c_int, [_vp, C.c_int, C.POINTER(C.c_double)]),
    ("mgcfd_rank_cycles", C.c_int, [_vp, C.c_int, C.POINTER(C.c_double)]),
    ("mgcfd_rank_ipc_detach", C.c_int, [_vp, C.c_int]),
    ("mgcfd_group_sweeps", C.c_int, [_vp, C.c_int, C.c_int]),
    ("mgcfd_group_sweeps_rms", C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    ("mgcfd_group_rms", C.c_int, [_vp, C.c_int, C.POINTER(C.c_double)]),
    ("mgcfd_group_synchronize", C.c_int, [_vp]),
]
EXPORTED_SYMBOLS = tuple(name for name, _, _ in _SIGNATURES)

_lib: Optional[C.CDLL] = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Load libmgcfd_hip.so and type every entry point.  Raises if the library is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p} not found: build it with `make -C {CSRC_DIR}` "
                                f"(or __graft_entry__.build()); there is no CPU fallback")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64, and a second runtime
    # brought up after the system one finds "no ROCm-capable device".  With torch loaded first this library binds to
    # the same (already loaded) runtime, so do that here when torch is installed — a process that never imports
    # torch (the C++ driver, a plain ctypes user) runs on the system runtime alone.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(p)
    for name, res, args in _SIGNATURES:
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _check(lib, rc: int):
    if rc != 0:
        raise MgcfdError(rc, (lib.mgcfd_last_error() or b"").decode(errors="replace"))


def _ptr(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_vp)


class Mesh:
    """Multigrid input parsed by the library's own readers (the reference's file formats)."""

    def __init__(self, input_dat: str, directory: str = "", duplicate: int = 1, legacy_ordering: bool = False):
        self.lib = load_library()
        h = _vp()
        _check(self.lib, self.lib.mgcfd_mesh_load_ex(input_dat.encode(), directory.encode(), duplicate,
                                                      1 if legacy_ordering else 0, C.byref(h)))
        self.handle = h

    @property
    def num_levels(self) -> int:
        return self.lib.mgcfd_mesh_num_levels(self.handle)

    @property
    def variant(self) -> int:
        return self.lib.mgcfd_mesh_variant(self.handle)

    @property
    def size(self) -> int:
        return self.lib.mgcfd_mesh_size(self.handle)

    def level(self, l: int) -> dict:
        d = LevelDesc()
        _check(self.lib, self.lib.mgcfd_mesh_level(self.handle, l, C.byref(d)))

        def view(addr, n, dt):
            if not addr or n == 0:
                return np.zeros(0, dtype=dt)
            buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(addr)
            return np.frombuffer(buf, dtype=dt, count=n).copy()

        return {"nel": d.nel, "n_edges": d.n_edges, "n_internal": d.n_internal, "n_boundary": d.n_boundary,
                "n_wall": d.n_wall, "internal_start": d.internal_start, "boundary_start": d.boundary_start,
                "wall_start": d.wall_start, "volumes": view(d.volumes, d.nel, np.float64),
                "coords": view(d.coords, d.nel * 3, np.float64).reshape(-1, 3),
                "edges": view(d.edges, d.n_edges, EDGE_DTYPE), "mg_map": view(d.mg_map, d.mgc, np.int64)}

    def close(self):
        if getattr(self, "handle", None):
            self.lib.mgcfd_mesh_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _level_descs(levels: Sequence[dict]):
    """LevelDesc array over in-memory level dicts + the arrays that must stay alive while it is used."""
    descs = (LevelDesc * len(levels))()
    keep = []
    for l, L in enumerate(levels):
        vol = np.ascontiguousarray(L["volumes"], dtype=np.float64)
        edges = np.ascontiguousarray(L["edges"], dtype=EDGE_DTYPE)
        crd = None if L.get("coords") is None else np.ascontiguousarray(L["coords"], dtype=np.float64)
        mp = None if L.get("mg_map") is None else np.ascontiguousarray(L["mg_map"], dtype=np.int64)
        keep += [vol, edges, crd, mp]
        d = descs[l]
        d.nel = int(L["nel"])
        d.n_edges = len(edges)
        d.n_internal, d.n_boundary, d.n_wall = int(L["n_internal"]), int(L["n_boundary"]), int(L["n_wall"])
        d.internal_start = int(L.get("internal_start", 0))
        d.boundary_start = int(L.get("boundary_start", d.n_internal))
        d.wall_start = int(L.get("wall_start", d.n_internal + d.n_boundary))
        d.volumes = _ptr(vol)
        d.coords = _ptr(crd) if crd is not None else None
        d.edges = _ptr(edges)
        d.mg_map = _ptr(mp) if mp is not None else None
        d.mgc = len(mp) if mp is not None else 0
    return descs, keep


def plan_audit(levels: Sequence[dict], mesh_variant: int, n_owned=None, order_keys=None) -> str:
    """Host only (no GPU): build the gather plans mgcfd_create would build for `levels` and check every index the kernels
    form from them against the size of what it indexes (mgcfd_plan_audit).  Returns "" when all is in range, else the report."""
    lib = load_library()
    descs, keep = _level_descs(levels)
    owned = None if n_owned is None else (_i64 * len(levels))(*[int(v) for v in n_owned])
    kp = None
    if order_keys is not None:
        keys = [None if k is None else np.ascontiguousarray(k, dtype=np.int64) for k in order_keys]
        keep += keys
        kp = (C.POINTER(_i64) * len(levels))(*[C.cast(_ptr(k), C.POINTER(_i64)) if k is not None else C.POINTER(_i64)() for k in keys])
    buf = C.create_string_buffer(1 << 16)
    rc = lib.mgcfd_plan_audit(descs, len(levels), mesh_variant, owned, kp, buf, len(buf))
    if rc not in (0, 1):
        _check(lib, rc)
    return buf.value.decode()


class Solver:
    """Device-resident solver.  Construct from a :class:`Mesh`, from a generated
    :class:`~mgcfd.meshgen.MultigridMesh`, or from raw per-level arrays."""

    def __init__(self, handle, lib):
        self.handle = handle
        self.lib = lib
        self._keep = None

    # ---- constructors ----
    @classmethod
    def from_mesh(cls, mesh: Mesh, device: int = 0) -> "Solver":
        lib = load_library()
        h = _vp()
        _check(lib, lib.mgcfd_create_from_mesh(mesh.handle, device, C.byref(h)))
        return cls(h, lib)

    @classmethod
    def from_arrays(cls, levels: Sequence[dict], mesh_variant: int, device: int = 0, n_owned=None, order_keys=None) -> "Solver":
        """levels[l] = dict(nel, volumes, coords|None, edges[EDGE_DTYPE], n_internal, n_boundary, n_wall,
        mg_map|None) — the reference's read_grid()/read_mg_connectivity() outputs.  n_owned / order_keys: a partitioned
        level or hierarchy (mgcfd_create_partitioned / _mg)."""
        lib = load_library()
        descs, keep = _level_descs(levels)
        h = _vp()
        if n_owned is None:
            _check(lib, lib.mgcfd_create(descs, len(levels), mesh_variant, device, C.byref(h)))
        elif order_keys is None:
            owned = (_i64 * len(levels))(*[int(v) for v in n_owned])
            _check(lib, lib.mgcfd_create_partitioned(descs, len(levels), mesh_variant, device, owned, C.byref(h)))
        else:
            owned = (_i64 * len(levels))(*[int(v) for v in n_owned])
            keys = [None if k is None else np.ascontiguousarray(k, dtype=np.int64) for k in order_keys]
            keep += keys
            kp = (C.POINTER(_i64) * len(levels))(*[C.cast(_ptr(k), C.POINTER(_i64)) if k is not None else C.POINTER(_i64)() for k in keys])
            _check(lib, lib.mgcfd_create_partitioned_mg(descs, len(levels), mesh_variant, device, owned, kp, C.byref(h)))
        return cls(h, lib)

    @classmethod
    def from_generated(cls, mg: MultigridMesh, device: int = 0) -> "Solver":
        return cls.from_arrays(generated_to_levels(mg), mg.mesh_variant, device)

    # ---- plumbing ----
    def _c(self, rc):
        _check(self.lib, rc)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.mgcfd_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name: str, value: int):
        self._c(self.lib.mgcfd_set_option(self.handle, OPT[name], int(value)))

    def array_devptr(self, l: int, name: str):
        """(device address, element count) of a node array as the library holds it ([ncols][stride] fp64,
        library numbering); see mgcfd_array_devptr."""
        p, n = _vp(), C.c_int64()
        self._c(self.lib.mgcfd_array_devptr(self.handle, l, ARR[name], C.byref(p), C.byref(n)))
        return p.value, n.value

    def accept_restricted(self, fine: int, dev_ptr: int):
        """Take the next-coarser level's restricted variables computed by another solver (one level per rank)."""
        self._c(self.lib.mgcfd_accept_restricted(self.handle, fine, C.c_void_p(dev_ptr)))

    def array_written(self, l: int, name: str):
        self._c(self.lib.mgcfd_array_written(self.handle, l, ARR[name]))

    def invalid_state_location(self):
        """(original cell id, 0-based cycle or -1) of the last invalid state a run reported."""
        cell, cycle = C.c_int64(-1), C.c_int(-1)
        self._c(self.lib.mgcfd_invalid_state_location(self.handle, C.byref(cell), C.byref(cycle)))
        return cell.value, cycle.value

    def tiling(self, l: int) -> dict:
        """How level ``l`` was cut into LDS tiles (mgcfd_level_tiling)."""
        out = (C.c_int64 * 10)()
        self._c(self.lib.mgcfd_level_tiling(self.handle, l, out))
        keys = ("tiles", "halo_nodes", "halo_max", "halo_capacity", "overflow_refs", "row_entries", "padding_entries", "coordinate_boxes",
                "list_entries", "loop_rows")
        return dict(zip(keys, (int(v) for v in out)))

    def has_half_rows(self, l: int) -> bool:
        yes = C.c_int(0)
        self._c(self.lib.mgcfd_level_has_half_rows(self.handle, l, C.byref(yes)))
        return bool(yes.value)

    def has_order_free(self, l: int) -> bool:
        yes = C.c_int(0)
        self._c(self.lib.mgcfd_level_has_order_free(self.handle, l, C.byref(yes)))
        return bool(yes.value)

    def has_edge_once(self, l: int) -> bool:
        v = C.c_int()
        self._c(self.lib.mgcfd_level_has_edge_once(self.handle, l, C.byref(v)))
        return bool(v.value)

    def get_option(self, name: str) -> int:
        v = C.c_int()
        self._c(self.lib.mgcfd_get_option(self.handle, OPT[name], C.byref(v)))
        return v.value

    def set_stream(self, stream_handle: Optional[int]):
        """Run this solver's work on an existing HIP stream (e.g. torch.cuda.Stream().cuda_stream);
        None restores the solver's own non-blocking stream.  Handle 0 is the legacy default stream,
        which the library cannot share (its own stream does not synchronise with it): refused, so that
        work torch enqueues (collectives, copies) is never silently unordered with the kernels."""
        if stream_handle is not None and int(stream_handle) == 0:
            raise ValueError("stream handle 0 is the legacy default stream; make a torch.cuda.Stream() current "
                             "(torch.cuda.set_stream) and pass its .cuda_stream, or pass None for the solver's own stream")
        self._c(self.lib.mgcfd_set_stream(self.handle, _vp(stream_handle) if stream_handle is not None else None))
        self._stream_handle = int(stream_handle) if stream_handle is not None else None

    def synchronize(self):
        self._c(self.lib.mgcfd_synchronize(self.handle))

    @property
    def num_levels(self) -> int:
        return self.lib.mgcfd_num_levels(self.handle)

    def nel(self, l: int) -> int:
        return self.lib.mgcfd_level_nel(self.handle, l)

    def num_internal_edges(self, l: int) -> int:
        return self.lib.mgcfd_level_num_internal_edges(self.handle, l)

    def far_field(self) -> np.ndarray:
        out = np.zeros(17)
        self._c(self.lib.mgcfd_get_far_field(self.handle, _ptr(out)))
        return out

    # ---- the reference's kernel set ----
    def copy_old_variables(self, l): self._c(self.lib.mgcfd_copy_old_variables(self.handle, l))
    def compute_step_factor(self, l): self._c(self.lib.mgcfd_compute_step_factor(self.handle, l))
    def compute_flux_edge(self, l): self._c(self.lib.mgcfd_compute_flux_edge(self.handle, l))
    def compute_boundary_flux_edge(self, l): self._c(self.lib.mgcfd_compute_boundary_flux_edge(self.handle, l))
    def compute_wall_flux_edge(self, l): self._c(self.lib.mgcfd_compute_wall_flux_edge(self.handle, l))
    def compute_fluxes(self, l): self._c(self.lib.mgcfd_compute_fluxes(self.handle, l))
    def time_step(self, l, j): self._c(self.lib.mgcfd_time_step(self.handle, l, j))
    def zero_fluxes(self, l): self._c(self.lib.mgcfd_zero_fluxes(self.handle, l))
    def indirect_rw(self, l): self._c(self.lib.mgcfd_indirect_rw(self.handle, l))
    def residual(self, l): self._c(self.lib.mgcfd_residual(self.handle, l))
    def restrict(self, fine): self._c(self.lib.mgcfd_restrict(self.handle, fine))
    def prolong(self, fine): self._c(self.lib.mgcfd_prolong(self.handle, fine))

    def calc_rms(self, l) -> float:
        v = C.c_double()
        self._c(self.lib.mgcfd_calc_rms(self.handle, l, C.byref(v)))
        return v.value

    def check_for_invalid_variables(self, l):
        bad = _i64(-1)
        rc = self.lib.mgcfd_check_for_invalid_variables(self.handle, l, C.byref(bad))
        return rc, bad.value

    def pending_invalid_state(self):
        """(code, cell) of what the checks inside the launches issued so far found; does not look at the current state."""
        bad = _i64(-1)
        rc = self.lib.mgcfd_pending_invalid_state(self.handle, C.byref(bad))
        return rc, bad.value

    def smooth(self, l: int, sweeps: int = 1):
        self._c(self.lib.mgcfd_smooth(self.handle, l, sweeps))

    def run_cycles(self, cycles: int) -> np.ndarray:
        rms = np.zeros(max(cycles, 1))
        self._c(self.lib.mgcfd_run_cycles(self.handle, cycles, _ptr(rms)))
        return rms[:cycles]

    # ---- state ----
    def get(self, l: int, name: str) -> np.ndarray:
        ncols = 1 if name in ("step_factors", "volumes") else NVAR
        out = np.zeros(self.nel(l) * ncols)
        self._c(self.lib.mgcfd_get_array(self.handle, l, ARR[name], _ptr(out)))
        return out.reshape(-1, ncols) if ncols > 1 else out

    def set(self, l: int, name: str, values: np.ndarray):
        a = np.ascontiguousarray(values, dtype=np.float64).ravel()
        ncols = 1 if name in ("step_factors", "volumes") else NVAR
        assert a.size == self.nel(l) * ncols
        self._c(self.lib.mgcfd_set_array(self.handle, l, ARR[name], _ptr(a)))

    def get_edges(self, l: int, n_edges: int) -> np.ndarray:
        out = np.zeros(n_edges, dtype=EDGE_DTYPE)
        self._c(self.lib.mgcfd_get_edges(self.handle, l, _ptr(out)))
        return out

    # ---- monitoring ----
    def loop_iters(self, l: int) -> dict:
        out = np.zeros(len(LOOPS), dtype=np.int64)
        self._c(self.lib.mgcfd_get_loop_iters(self.handle, l, _ptr(out)))
        return dict(zip(LOOPS, out.tolist()))

    def loop_times(self, l: int) -> dict:
        out = np.zeros(len(LOOPS))
        self._c(self.lib.mgcfd_get_loop_times(self.handle, l, _ptr(out)))
        return dict(zip(LOOPS, out.tolist()))

    def reset_monitoring(self):
        self._c(self.lib.mgcfd_reset_monitoring(self.handle))

    def flux_kernel_time(self, l: int):
        t = C.c_double()
        n = _i64()
        self._c(self.lib.mgcfd_get_flux_kernel_time(self.handle, l, C.byref(t), C.byref(n)))
        return t.value, n.value

    def bench_flux(self, l: int, launches: int) -> float:
        t = C.c_double()
        self._c(self.lib.mgcfd_bench_flux(self.handle, l, launches, C.byref(t)))
        return t.value

    def bench_stream_ceiling(self, l: int, launches: int) -> float:
        """Mean seconds per launch of a tile-shaped stream of exactly the flux launch's algorithmic bytes (mgcfd_bench_stream_ceiling)."""
        t = C.c_double(0.0)
        self._c(self.lib.mgcfd_bench_stream_ceiling(self.handle, l, launches, C.byref(t)))
        return t.value

    def bench_indirect_rw(self, l: int, launches: int) -> float:
        t = C.c_double()
        self._c(self.lib.mgcfd_bench_indirect_rw(self.handle, l, launches, C.byref(t)))
        return t.value

    # ---- halo exchange of a partitioned level ----
    def halo_plan(self, l: int, node_ids) -> int:
        ids = np.ascontiguousarray(node_ids, dtype=np.int64)
        plan = C.c_int(-1)
        self._c(self.lib.mgcfd_halo_plan(self.handle, l, len(ids), _ptr(ids) if len(ids) else None, C.byref(plan)))
        return plan.value

    def halo_pack(self, l: int, plan: int, name: str, dev_ptr: int):
        self._c(self.lib.mgcfd_halo_pack(self.handle, l, plan, ARR[name], _vp(dev_ptr)))

    def halo_unpack(self, l: int, plan: int, name: str, dev_ptr: int):
        self._c(self.lib.mgcfd_halo_unpack(self.handle, l, plan, ARR[name], _vp(dev_ptr)))

    # ---- multi-GPU hooks ----
    def step_factor_local(self, l): self._c(self.lib.mgcfd_step_factor_local(self.handle, l))
    def step_factor_apply(self, l): self._c(self.lib.mgcfd_step_factor_apply(self.handle, l))

    def sweep_begin(self, l): self._c(self.lib.mgcfd_sweep_begin(self.handle, l))
    def sweep_flux0(self, l): self._c(self.lib.mgcfd_sweep_flux0(self.handle, l))
    def sweep_end(self, l): self._c(self.lib.mgcfd_sweep_end(self.handle, l))

    def step_factor_min_devptr(self, l) -> int:
        p = _vp()
        self._c(self.lib.mgcfd_step_factor_min_devptr(self.handle, l, C.byref(p)))
        return p.value

    def step_factor_partials_devptr(self, l):
        p, n = _vp(), C.c_int()
        self._c(self.lib.mgcfd_step_factor_partials_devptr(self.handle, l, C.byref(p), C.byref(n)))
        return p.value, n.value

    def sweep_stage(self, l, j, partials=True): self._c(self.lib.mgcfd_sweep_stage(self.handle, l, j, 1 if partials else 0))
    def sweep_begin_partials(self, l): self._c(self.lib.mgcfd_sweep_begin_partials(self.handle, l))
    def sweep_end_partials(self, l): self._c(self.lib.mgcfd_sweep_end_partials(self.handle, l))

    # ---- a rank of a partitioned level, the sweep loop inside the library (include/mgcfd.h "Multi-GPU in the C++ host") ----
    def rank_set_halo(self, l: int, part):
        """part: mgcfd.partition.LevelPart — its send / recv lists (local ids per peer, ascending global id)."""
        peers = sorted(set(part.send) | set(part.recv))
        n = len(peers)
        keep = [np.ascontiguousarray(part.send.get(p, np.zeros(0, np.int64)), dtype=np.int64) for p in peers] + \
               [np.ascontiguousarray(part.recv.get(p, np.zeros(0, np.int64)), dtype=np.int64) for p in peers]
        pa = (C.c_int * max(n, 1))(*peers)
        sc = (_i64 * max(n, 1))(*[len(a) for a in keep[:n]])
        rc = (_i64 * max(n, 1))(*[len(a) for a in keep[n:]])
        sp = (_vp * max(n, 1))(*[a.ctypes.data for a in keep[:n]])
        rp = (_vp * max(n, 1))(*[a.ctypes.data for a in keep[n:]])
        self._c(self.lib.mgcfd_rank_set_halo(self.handle, l, n, pa, sc, sp, rc, rp))

    def rank_halo_info(self, l: int) -> dict:
        out = (_i64 * 4)()
        self._c(self.lib.mgcfd_rank_halo_info(self.handle, l, out))
        return dict(zip(("boundary_tiles", "interior_tiles", "nodes_sent", "nodes_received"), [int(v) for v in out]))

    def rank_graph_status(self, l: int) -> dict:
        """MGCFD_OPT_GRAPH on an RCCL rank: graphs instantiated, whether a capture was refused, sweeps replayed (mgcfd_rank_graph_status)."""
        out = (_i64 * 3)()
        self._c(self.lib.mgcfd_rank_graph_status(self.handle, l, out))
        return {"graphs": int(out[0]), "capture_refused": bool(out[1]), "sweeps_replayed": int(out[2])}

    def rank_info(self) -> dict:
        """What this solver is a rank of, as the library sees it (mgcfd_rank_info)."""
        out = (C.c_int * 4)()
        self._c(self.lib.mgcfd_rank_info(self.handle, out))
        return {"rank": out[0], "ranks": out[1], "transport": ("none", "rccl", "in-process group", "plain (HIP IPC messages)")[out[2]], "rccl_comm_count": out[3]}

    def rank_attach_rccl(self, rank: int, world: int, unique_id: bytes):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._c(self.lib.mgcfd_rank_attach_rccl(self.handle, rank, world, buf))

    def rank_detach(self): self._c(self.lib.mgcfd_rank_detach(self.handle))
    def rank_attach_plain(self, rank: int, world: int): self._c(self.lib.mgcfd_rank_attach_plain(self.handle, rank, world))

    def rank_ipc_export(self, l: int) -> bytes:
        """HIP IPC handles of this rank's state buffers and flag words + its ghost list: for the neighbouring ranks' rank_ipc_attach."""
        n = _i64(0)
        self._c(self.lib.mgcfd_rank_ipc_export_size(self.handle, l, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        self._c(self.lib.mgcfd_rank_ipc_export(self.handle, l, buf))
        return buf.raw

    def rank_ipc_attach(self, l: int, exports):
        """exports: what other ranks exported (any order; the own one is skipped): at least every neighbour's — with every
        rank's the time-step all-reduce goes through the flags as well."""
        keep = [C.create_string_buffer(e, len(e)) for e in exports]
        arr = (_vp * max(len(keep), 1))(*[C.cast(b, _vp) for b in keep])
        self._c(self.lib.mgcfd_rank_ipc_attach(self.handle, l, len(keep), arr))

    def rank_ipc_detach(self, l: int): self._c(self.lib.mgcfd_rank_ipc_detach(self.handle, l))

    def rank_ipc_status(self, l: int) -> int:
        n = C.c_int(0)
        self._c(self.lib.mgcfd_rank_ipc_status(self.handle, l, C.byref(n)))
        return n.value
    def rank_exchange(self, l: int): self._c(self.lib.mgcfd_rank_exchange(self.handle, l))
    def rank_sweeps(self, l: int, sweeps: int = 1): self._c(self.lib.mgcfd_rank_sweeps(self.handle, l, sweeps))

    def rank_cycles(self, cycles: int, rms: bool = True) -> np.ndarray:
        """V-cycles of this rank's share of a partitioned hierarchy over RCCL (mgcfd_rank_cycles); the level-0 RMS of each cycle."""
        out = np.zeros(max(cycles, 1), dtype=np.float64)
        self._c(self.lib.mgcfd_rank_cycles(self.handle, cycles, out.ctypes.data_as(C.POINTER(C.c_double)) if rms else None))
        return out[:cycles]

    def rank_residual_sumsq(self, l: int) -> float:
        v = C.c_double()
        self._c(self.lib.mgcfd_rank_residual_sumsq(self.handle, l, C.byref(v)))
        return v.value

    def residual_sumsq_devptr(self, l) -> int:
        p = _vp()
        self._c(self.lib.mgcfd_residual_sumsq(self.handle, l, C.byref(p)))
        return p.value


def generated_to_levels(mg: MultigridMesh) -> List[dict]:
    """Turn a generated mesh into read_grid()-shaped per-level dicts without touching disk."""
    out = []
    for lvl in mg.levels:
        edges, ni, nb, nw = to_edge_arrays(lvl, mg.mesh_variant)
        out.append({"nel": lvl.nel, "volumes": lvl.volumes, "coords": lvl.coords, "edges": edges,
                    "n_internal": ni, "n_boundary": nb, "n_wall": nw, "mg_map": lvl.mg_map})
    return out


def rccl_unique_id() -> bytes:
    """ncclGetUniqueId through the library (rank 0 calls it; the launcher hands the 128 bytes to every rank)."""
    lib = load_library()
    buf = C.create_string_buffer(128)
    _check(lib, lib.mgcfd_rccl_unique_id(buf))
    return buf.raw


class Group:
    """The solvers of THIS process as the ranks of one partitioned level (mgcfd_group_*): solvers[r] = rank r."""

    def __init__(self, solvers):
        self.lib = load_library()
        self.solvers = list(solvers)
        arr = (_vp * len(self.solvers))(*[s.handle for s in self.solvers])
        h = _vp()
        _check(self.lib, self.lib.mgcfd_group_create(len(self.solvers), arr, C.byref(h)))
        self.handle = h

    def exchange(self, l: int = 0): _check(self.lib, self.lib.mgcfd_group_exchange(self.handle, l))
    def sweeps(self, l: int = 0, n: int = 1): _check(self.lib, self.lib.mgcfd_group_sweeps(self.handle, l, n))

    def sweeps_rms(self, l: int = 0, n: int = 1) -> np.ndarray:
        """n sweeps, the RMS after each (read back once at the end)."""
        out = np.zeros(max(n, 1), dtype=np.float64)
        _check(self.lib, self.lib.mgcfd_group_sweeps_rms(self.handle, l, n, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out[:n]
    def synchronize(self): _check(self.lib, self.lib.mgcfd_group_synchronize(self.handle))

    def cycles(self, n: int = 1, rms: bool = True) -> np.ndarray:
        """n V-cycles of a partitioned hierarchy (mgcfd_group_cycles); the level-0 RMS of each cycle."""
        out = np.zeros(max(n, 1), dtype=np.float64)
        _check(self.lib, self.lib.mgcfd_group_cycles(self.handle, n, out.ctypes.data_as(C.POINTER(C.c_double)) if rms else None))
        return out[:n]

    def rms(self, l: int = 0) -> float:
        v = C.c_double()
        _check(self.lib, self.lib.mgcfd_group_rms(self.handle, l, C.byref(v)))
        return v.value

    def close(self):
        if self.handle:
            self.lib.mgcfd_group_destroy(self.handle)
            self.handle = None
