#!/usr/bin/env python3
"""Per-phase wall-clock ticks inside k_flux_half (diagnostic build with -DMGCFD_PHASES, never shipped).
build (here):  python tools/phase_half.py build      -> csrc/build/exp/libmgcfd_hip_phases.so (travels to the GPU box)
run (GPU box): python tools/phase_half.py run [variant=32] [lattice=67]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mg-cfd-app-plain_amd", "csrc")
OUT = os.path.join(CSRC, "build", "exp")
LIB = os.environ.get("PH_LIB") or os.path.join(OUT, "libmgcfd_hip_phases.so")

def build():
    os.makedirs(OUT, exist_ok=True)
    objs = []
    for ns, contract, extra in (("exact", "off", ["-DMGCFD_PHASE_EXPORT"]), ("fast", "fast", [])):
        o = os.path.join(OUT, f"k_phases_{ns}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-Wno-unused-result",
                               "-mllvm", "-amdgpu-kernarg-preload-count=16", f"-ffp-contract={contract}", f"-DMGCFD_KERNEL_NS={ns}", "-DMGCFD_PHASES", *os.environ.get("PH_DEFS", "").split(),
                               f"-I{ROOT}/include", f"-I{CSRC}", "-c", os.path.join(CSRC, "kernels.hip"), "-o", o] + extra)
        objs.append(o)
    host = [os.path.join(CSRC, "build", x) for x in ("solver.o", "mesh.o", "preprocess.o")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + host)
    print("built", LIB)

def run():
    os.environ["MGCFD_LIB"] = LIB
    sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
    import numpy as np
    import bench, mgcfd
    variant = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    lattice = int(sys.argv[3]) if len(sys.argv) > 3 else 67
    mg, levels = bench.build_workload(lattice)
    s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    s.set(0, "variables", bench.perturbed_state(s.nel(0), s.far_field()[:5]))
    s.set_option("flux_variant", variant)
    lib = C.CDLL(LIB)
    buf = (C.c_ulonglong * (4096 * 8))()
    if os.environ.get("PH_PROBE"):        # time the indirect_rw probe instead (ablation builds)
        print(f"indirect_rw through the tiles: {s.bench_indirect_rw(0, 200) * 1e6:.2f} us; flux kernel {s.bench_flux(0, 200) * 1e6:.2f} us")
        return
    s.bench_flux(0, 20)
    lib.mgcfd_debug_phases(buf, 1)
    t = s.bench_flux(0, 200)
    lib.mgcfd_debug_phases(buf, 1)
    a = np.ctypeslib.as_array(buf).reshape(4096, 8).astype(np.float64)
    n = a[:, 7].sum()
    tot = a.sum(0)
    names = ["stage + barrier", "half rows", "barrier (records dead)", "hand-over + barrier", "ordered adds", "boundary + store"]
    print(f"variant {variant} lattice {lattice}: kernel avg {t*1e6:.2f} us, {n:.0f} workgroups timed")
    for k, nm in enumerate(names):
        print(f"  {nm:24s} {tot[k] / n / 100.0:7.3f} us per workgroup")
    print(f"  {'total':24s} {sum(tot[:6]) / n / 100.0:7.3f} us per workgroup")

if __name__ == "__main__":
    (build if sys.argv[1] == "build" else run)()
