"""ctypes loader for the CPU oracle (libmgcfd_oracle.so) and, when present, the real
reference behind our harness (oracle/_ref/libmgcfd_ref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s cpu_baseline leg.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NVAR = 5

EDGE_DTYPE = np.dtype([("a", "<i8"), ("b", "<i8"), ("x", "<f8"), ("y", "<f8"), ("z", "<f8")])

_dp = C.POINTER(C.c_double)
_lp = C.POINTER(C.c_int64)
_vp = C.c_void_p


class OraFarfield(C.Structure):
    _fields_ = [("var", C.c_double * 5), ("fc_mx", C.c_double * 3), ("fc_my", C.c_double * 3),
                ("fc_mz", C.c_double * 3), ("fc_de", C.c_double * 3)]


class OraLevel(C.Structure):
    _fields_ = [("nel", C.c_int64), ("n_edges", C.c_int64), ("n_internal", C.c_int64),
                ("n_boundary", C.c_int64), ("n_wall", C.c_int64), ("internal_start", C.c_int64),
                ("boundary_start", C.c_int64), ("wall_start", C.c_int64),
                ("volumes", _vp), ("coords", _vp), ("edges", _vp), ("mg_map", _vp), ("mgc", C.c_int64),
                ("variables", _vp), ("old_variables", _vp), ("residuals", _vp), ("fluxes", _vp),
                ("step_factors", _vp)]


class OraIters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("flux", "update", "compute_step", "time_step", "restrict_",
                                         "prolong", "indirect_rw")]


def build(native: bool = False) -> str:
    """(Re)build the oracle shared objects with make; returns the path of the requested one."""
    subprocess.run(["make", "-s", "-C", HERE, "all"], check=True)
    return os.path.join(HERE, "libmgcfd_oracle_native.so" if native else "libmgcfd_oracle.so")


def ptr(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_vp)


_cache = {}


def load(native: bool = False) -> C.CDLL:
    key = "native" if native else "canon"
    if key in _cache:
        return _cache[key]
    path = os.path.join(HERE, "libmgcfd_oracle_native.so" if native else "libmgcfd_oracle.so")
    src = os.path.join(HERE, "mgcfd_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        build(native)
    lib = C.CDLL(path)
    i64, dbl, i32 = C.c_int64, C.c_double, C.c_int
    lib.ora_far_field.argtypes = [C.POINTER(OraFarfield)]
    lib.ora_initialize_variables.argtypes = [i64, _vp, C.POINTER(OraFarfield)]
    for name in ("ora_compute_flux_edge", "ora_compute_boundary_flux_edge", "ora_indirect_rw"):
        getattr(lib, name).argtypes = [i64, i64, _vp, _vp, _vp]
    lib.ora_compute_wall_flux_edge.argtypes = [i64, i64, _vp, _vp, _vp, C.POINTER(OraFarfield)]
    lib.ora_compute_step_factor.argtypes = [i64, _vp, _vp, _vp]
    lib.ora_compute_step_factor_legacy.argtypes = [i64, _vp, _vp, _vp]
    lib.ora_time_step.argtypes = [i32, i64, _vp, _vp, _vp, _vp]
    lib.ora_zero_fluxes.argtypes = [i64, _vp]
    lib.ora_residual.argtypes = [i64, _vp, _vp, _vp]
    lib.ora_calc_rms.argtypes = [i64, _vp]
    lib.ora_calc_rms.restype = dbl
    lib.ora_adjust_ewt.argtypes = [_vp, i64, _vp]
    lib.ora_dampen_ewt.argtypes = [i64, _vp, dbl]
    lib.ora_check_for_invalid_variables.argtypes = [_vp, i64, _lp]
    lib.ora_check_for_invalid_variables.restype = i32
    lib.ora_identify_differences.argtypes = [_vp, _vp, i64, i32]
    lib.ora_identify_differences.restype = i64
    lib.ora_mg_restrict.argtypes = [_vp, _vp, i64, _vp, _vp, i64]
    lib.ora_prolong_residuals_interpolate_proper.argtypes = [_vp, i64, _vp, _vp, _vp, i64, _vp, _vp, _vp]
    lib.ora_read_grid.argtypes = [C.c_char_p, i32, i32, C.POINTER(OraLevel)]
    lib.ora_read_grid.restype = i32
    lib.ora_read_mg_connectivity.argtypes = [C.c_char_p, C.POINTER(_vp), _lp]
    lib.ora_read_mg_connectivity.restype = i32
    lib.ora_sort_edges_legacy.argtypes = [C.POINTER(OraLevel)]
    lib.ora_duplicate_mesh.argtypes = [C.POINTER(OraLevel), i32, i64]
    lib.ora_alloc_state.argtypes = [C.POINTER(OraLevel)]
    lib.ora_free_level.argtypes = [C.POINTER(OraLevel)]
    lib.ora_solve.argtypes = [C.POINTER(OraLevel), i32, i32, i32, i32, _vp, C.POINTER(OraIters)]
    lib.ora_solve.restype = i32
    _cache[key] = lib
    return lib


def farfield() -> OraFarfield:
    ff = OraFarfield()
    load().ora_far_field(C.byref(ff))
    return ff


def _view(addr, n, dtype):
    if not addr or n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype, count=n)


class OracleCase:
    """A multigrid input held by the oracle: read from the reference-format files (or built
    from in-memory arrays) and solved with ``ora_solve``."""

    def __init__(self, mesh_variant: int):
        self.lib = load()
        self.mesh_variant = mesh_variant
        self.levels = None
        self.nlevels = 0

    @classmethod
    def from_files(cls, level_paths, map_paths, mesh_variant: int, duplicate: int = 1, legacy_ordering: bool = False,
                   coords_as_reference: bool = False) -> "OracleCase":
        """coords_as_reference: read <mesh>.coords only when there is more than one level, exactly as the reference does
        (src/Base/io.cpp:49-54,77-81) — a single-level m6wing / la_cascade / rotor37 input then has all-zero coordinates,
        adjust_ewt divides by a zero distance and the run aborts with NaN, as the reference binary's does.  Default: also
        read the file when it exists (what the library does), so such inputs can be used."""
        self = cls(mesh_variant)
        n = len(level_paths)
        self.nlevels = n
        self.levels = (OraLevel * n)()
        for l, p in enumerate(level_paths):
            rc = self.lib.ora_read_grid(p.encode(), mesh_variant,
                                        1 if (n > 1 or (not coords_as_reference and os.path.exists(p + ".coords"))) else 0,
                                        C.byref(self.levels[l]))
            if rc:
                raise RuntimeError(f"ora_read_grid({p}) failed rc={rc}")
            if legacy_ordering:
                self.lib.ora_sort_edges_legacy(C.byref(self.levels[l]))
            if l < n - 1:
                m = _vp()
                mgc = C.c_int64()
                rc = self.lib.ora_read_mg_connectivity(map_paths[l].encode(), C.byref(m), C.byref(mgc))
                if rc:
                    raise RuntimeError(f"ora_read_mg_connectivity({map_paths[l]}) failed rc={rc}")
                self.levels[l].mg_map = m
                self.levels[l].mgc = mgc.value
        if duplicate > 1:
            nel_above = [self.levels[l + 1].nel if l < n - 1 else 0 for l in range(n)]
            for l in range(n):
                self.lib.ora_duplicate_mesh(C.byref(self.levels[l]), duplicate, nel_above[l])
        for l in range(n):
            self.lib.ora_alloc_state(C.byref(self.levels[l]))
        return self

    @classmethod
    def from_input_dat(cls, dat_path: str, duplicate: int = 1, legacy_ordering: bool = False,
                       coords_as_reference: bool = False) -> "OracleCase":
        info = parse_input_dat(dat_path)
        d = os.path.dirname(dat_path)
        return cls.from_files([os.path.join(d, p) for p in info["levels"]],
                              [os.path.join(d, p) for p in info["mg_mapping"]],
                              info["mesh_variant"], duplicate, legacy_ordering, coords_as_reference)

    # -- array views (owned by the C side) --
    def edges(self, l):
        L = self.levels[l]
        return _view(L.edges, L.n_edges, EDGE_DTYPE)

    def array(self, l, name):
        L = self.levels[l]
        n = {"volumes": L.nel, "step_factors": L.nel, "coords": L.nel * 3}.get(name, L.nel * NVAR)
        return _view(getattr(L, name), n, np.float64)

    def mg_map(self, l):
        L = self.levels[l]
        return _view(L.mg_map, L.mgc, np.int64)

    def solve(self, cycles: int, run_indirect_rw: bool = False):
        rms = np.zeros(max(cycles, 1))
        iters = (OraIters * self.nlevels)()
        rc = self.lib.ora_solve(self.levels, self.nlevels, self.mesh_variant, cycles,
                                1 if run_indirect_rw else 0, ptr(rms), iters)
        return rc, rms[:cycles], iters

    def close(self):
        if self.levels is not None:
            for l in range(self.nlevels):
                self.lib.ora_free_level(C.byref(self.levels[l]))
            self.levels = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def parse_input_dat(path: str) -> dict:
    """Minimal input.dat parser for the test side (format: src/Base/io_enhanced.cpp:407-579)."""
    codes = {"fvcorr": 0, "m6wing": 2, "la_cascade": 3, "rotor37": 4}
    out = {"levels": [], "mg_mapping": [], "size": 1}
    section = None
    with open(path) as f:
        for raw in f:
            line = raw.rstrip("\n")
            if line.startswith("#") or not line.strip():
                continue
            if line.startswith("["):
                section = line.strip()[1:-1]
                continue
            if "=" not in line:
                continue
            k, v = [s.strip() for s in line.split("=", 1)]
            if section in ("levels", "mg_mapping") and k.isdigit():
                lst = out[section]
                while len(lst) <= int(k):
                    lst.append("")
                lst[int(k)] = v
            elif k == "size":
                out["size"] = int(v)
            elif k == "num_levels":
                out["num_levels"] = int(v)
            elif k == "mesh_name":
                out["mesh_name"] = v
                out["mesh_variant"] = codes[v]
    return out


# ---------------------------------------------------------------------------------------
# The real reference, when oracle/_ref/ has been built (this container only)
# ---------------------------------------------------------------------------------------
REF_DIR = os.path.join(HERE, "_ref")
REF_LIB = os.path.join(REF_DIR, "libmgcfd_ref.so")
REF_BIN = os.path.join(REF_DIR, "euler3d_cpu_double_ref.b")
REF_BIN_LEGACY = os.path.join(REF_DIR, "euler3d_cpu_double_ref_legacy_ordering.b")


def have_reference() -> bool:
    return os.path.exists(REF_LIB) and os.path.exists(REF_BIN)


def load_reference() -> C.CDLL:
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    lib = C.CDLL(REF_LIB)
    l, d, i = C.c_long, C.c_double, C.c_int
    lib.ref_init.argtypes = [i, i]
    lib.ref_set_level.argtypes = [i]
    lib.ref_get_farfield.argtypes = [_vp]
    for name in ("ref_compute_flux_edge", "ref_compute_boundary_flux_edge", "ref_compute_wall_flux_edge",
                 "ref_indirect_rw"):
        getattr(lib, name).argtypes = [l, l, _vp, _vp, _vp]
    lib.ref_compute_step_factor.argtypes = [l, _vp, _vp, _vp]
    lib.ref_compute_step_factor_legacy.argtypes = [l, _vp, _vp, _vp]
    lib.ref_time_step.argtypes = [i, l, _vp, _vp, _vp, _vp]
    lib.ref_residual.argtypes = [l, _vp, _vp, _vp]
    lib.ref_calc_rms.argtypes = [l, _vp]
    lib.ref_calc_rms.restype = d
    lib.ref_adjust_ewt.argtypes = [_vp, l, _vp]
    lib.ref_dampen_ewt.argtypes = [l, _vp, d]
    lib.ref_mg_restrict.argtypes = [_vp, _vp, l, _vp, _vp, l]
    lib.ref_prolong_residuals_interpolate_proper.argtypes = [_vp, l, _vp, _vp, _vp, l, _vp, _vp, _vp]
    lib.ref_read_grid.argtypes = [C.c_char_p, _vp]
    lib.ref_grid_copy.argtypes = [_vp, _vp, _vp]
    return lib
