#!/usr/bin/env python3
"""Host-side cost of the drop-in's start-up, no GPU: reading an input directory (mgcfd_mesh_load) and building every level's
gather and transfer plans (mgcfd_plan_audit builds exactly what mgcfd_create builds).   python tools/host_setup_time.py DIR [repeats]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd"))
import mgcfd
from mgcfd import api
d = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lib = mgcfd.load_library()
best_r = best_p = 1e9
for _ in range(reps):
    t0 = time.perf_counter()
    m = mgcfd.Mesh("input.dat", d)
    t1 = time.perf_counter()
    descs = (api.LevelDesc * m.num_levels)()
    for l in range(m.num_levels):
        lib.mgcfd_mesh_level(m.handle, l, C.byref(descs[l]))
    buf = C.create_string_buffer(1 << 16)
    rc = lib.mgcfd_plan_audit(descs, m.num_levels, lib.mgcfd_mesh_variant(m.handle), None, None, buf, len(buf))
    t2 = time.perf_counter()
    best_r, best_p = min(best_r, t1 - t0), min(best_p, t2 - t1)
    del m
print(f"read {best_r:.3f} s   plans (+ audit) {best_p:.3f} s   audit: {'clean' if not buf.value else buf.value.decode()[:200]}  rc {rc}")
