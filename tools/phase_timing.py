#!/usr/bin/env python3
"""Per-phase clock counts inside k_flux_edge_once / k_flux_tile (debug build, never shipped).

build:  python tools/phase_timing.py build      (here; writes csrc/build/exp/libmgcfd_hip_phases.so)
run:    python tools/phase_timing.py run        (GPU box)
Thread 0 of every workgroup adds the s_memtime deltas between phase boundaries to a device
array; the host prints the mean per workgroup in microseconds (100 MHz wall clock)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mg-cfd-app-plain_amd", "csrc")
OUT = os.path.join(CSRC, "build", "exp")
LIB = os.path.join(OUT, "libmgcfd_hip_phases.so")

def rep(s, a, b):
    assert s.count(a) >= 1, a
    return s.replace(a, b, 1)

def build():
    src = open(os.environ.get("PH_SRC", os.path.join(CSRC, "kernels.hip"))).read()
    if os.environ.get("PH_TILE"):
        return build_tile(src)
    head = '''
__device__ unsigned long long g_phase[4096 * 8];
#define PH_MARK(k) do { if (threadIdx.x == 0) { unsigned long long now_ = wall_clock64(); g_phase[blockIdx.x * 8 + k] += now_ - ph_last_; ph_last_ = now_; } } while (0)
#define PH_BEGIN() unsigned long long ph_last_ = wall_clock64(); if (threadIdx.x == 0) g_phase[blockIdx.x * 8 + 7] += 1ull
'''
    i = src.index("template <bool LOADK, bool FUSE, bool ACC>\n__global__ void __launch_bounds__(kBlock, 3)\nk_flux_edge_once")
    src = src[:i] + head + src[i:]
    j = src.index("k_flux_edge_once", i)
    body = src[j:]
    body = rep(body, "    double min_dt = 0.0;\n", "    PH_BEGIN();\n    double min_dt = 0.0;\n")
    if os.environ.get("PH_FINE"):
        body = rep(body, "    const bool has_halo = hid >= 0;\n", "    asm volatile(\"\" :: \"v\"(hid));\n    PH_MARK(0);\n    const bool has_halo = hid >= 0;\n")
        body = rep(body, "        lds_store_record(tile, uint32_t(tid), make_nodeq(o0, o1, o2, o3, o4));", "        asm volatile(\"\" :: \"v\"(g4), \"v\"(o4), \"v\"(g0));\n        PH_MARK(1);\n        lds_store_record(tile, uint32_t(tid), make_nodeq(o0, o1, o2, o3, o4));")
        body = rep(body, "    const int32_t ovf0 = tile_ovf_ptr[t];\n    __syncthreads();\n", "    const int32_t ovf0 = tile_ovf_ptr[t];\n    PH_MARK(2);\n    __syncthreads();\n    PH_MARK(3);\n")
        src = src[:j] + body
        names_fine = True
    else:
      body = rep(body, "    __syncthreads();\n\n    // ---- phase 2", "    __syncthreads();\n    PH_MARK(0);\n\n    // ---- phase 2")
      body = rep(body, "    double *fb = reinterpret_cast<double *>(tile);\n    __syncthreads();\n", "    double *fb = reinterpret_cast<double *>(tile);\n    PH_MARK(1);\n    __syncthreads();\n    PH_MARK(2);\n")
      body = rep(body, "    __syncthreads();\n\n    // ---- phase 4", "    __syncthreads();\n    PH_MARK(3);\n\n    // ---- phase 4")
      body = rep(body, "    if ((classes & 6) && n_bnd > 0) {\n        // boundary faces need", "    PH_MARK(4);\n    if ((classes & 6) && n_bnd > 0) {\n        // boundary faces need")
      body = rep(body, "    finish_node<FUSE>(i, nel, stride, a0, a1, a2, a3, a4, fluxes, fs, min_dt, t);\n}", "    finish_node<FUSE>(i, nel, stride, a0, a1, a2, a3, a4, fluxes, fs, min_dt, t);\n    PH_MARK(5);\n}")
      src = src[:j] + body
    src += '''
#ifdef MGCFD_PHASE_EXPORT
extern "C" void mgcfd_debug_phases(unsigned long long *out, int reset)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(mgcfd::MGCFD_KERNEL_NS::g_phase), sizeof(unsigned long long) * 4096 * 8);
    if (reset) { static unsigned long long z[4096 * 8]; (void)hipMemcpyToSymbol(HIP_SYMBOL(mgcfd::MGCFD_KERNEL_NS::g_phase), z, sizeof(z)); }
}
#endif
'''
    os.makedirs(OUT, exist_ok=True)
    f = os.path.join(OUT, "kernels_phases.hip")
    open(f, "w").write(src)
    objs = []
    for ns, contract, extra in (("exact", "off", ["-DMGCFD_PHASE_EXPORT"]), ("fast", "fast", [])):
        o = os.path.join(OUT, f"k_phases_{ns}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math",
                               f"-ffp-contract={contract}", f"-DMGCFD_KERNEL_NS={ns}", f"-I{ROOT}/include", f"-I{CSRC}", "-c", f, "-o", o] + extra)
        objs.append(o)
    host = [os.path.join(CSRC, "build", x) for x in ("solver.o", "mesh.o", "preprocess.o")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + host)
    print("built", LIB)

HEAD = '''
__device__ unsigned long long g_phase[4096 * 8];
#define PH_MARK(k) do { if (threadIdx.x == 0) { unsigned long long now_ = wall_clock64(); g_phase[blockIdx.x * 8 + k] += now_ - ph_last_; ph_last_ = now_; } } while (0)
#define PH_BEGIN() unsigned long long ph_last_ = wall_clock64(); if (threadIdx.x == 0) g_phase[blockIdx.x * 8 + 7] += 1ull
'''
TAIL = '''
#ifdef MGCFD_PHASE_EXPORT
extern "C" void mgcfd_debug_phases(unsigned long long *out, int reset)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(mgcfd::MGCFD_KERNEL_NS::g_phase), sizeof(unsigned long long) * 4096 * 8);
    if (reset) { static unsigned long long z[4096 * 8]; (void)hipMemcpyToSymbol(HIP_SYMBOL(mgcfd::MGCFD_KERNEL_NS::g_phase), z, sizeof(z)); }
}
#endif
'''

def compile_lib(src, host_dir=None):
    os.makedirs(OUT, exist_ok=True)
    f = os.path.join(OUT, "kernels_phases.hip")
    open(f, "w").write(src)
    inc = os.environ.get("PH_INC", CSRC)
    objs = []
    for ns, contract, extra in (("exact", "off", ["-DMGCFD_PHASE_EXPORT"]), ("fast", "fast", [])):
        o = os.path.join(OUT, f"k_phases_{ns}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math",
                               f"-ffp-contract={contract}", f"-DMGCFD_KERNEL_NS={ns}", f"-I{ROOT}/include", f"-I{inc}", "-c", f, "-o", o] + extra)
        objs.append(o)
    hd = os.environ.get("PH_HOST", os.path.join(CSRC, "build"))
    host = [os.path.join(hd, x) for x in ("solver.o", "mesh.o", "preprocess.o")]
    lib = os.environ.get("PH_LIB", LIB)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + host)
    print("built", lib)

def build_tile(src):
    i = src.index("template <int MINW, bool LOADK, bool FUSE, bool ACC, int ROLE, bool TAIL>\n__global__ void __launch_bounds__(kBlock, MINW)\nk_flux_tile")
    src = src[:i] + HEAD + src[i:]
    j = src.index("k_flux_tile", i)
    body = src[j:]
    body = rep(body, "    double min_dt = 0.0;\n", "    PH_BEGIN();\n    double min_dt = 0.0;\n")
    body = rep(body, "    // ---- phase 2: incidence rows two at a time", "    PH_MARK(0);\n    // ---- phase 2: incidence rows two at a time")
    body = rep(body, "    if (TAIL && (classes & 1)) {\n", "    PH_MARK(1);\n    if (TAIL && (classes & 1)) {\n")
    body = rep(body, "    if ((classes & 6) && n_bnd > 0) {\n", "    PH_MARK(2);\n    if ((classes & 6) && n_bnd > 0) {\n")
    # only k_flux_tile's own text (up to the edge-once kernel's banner)
    end = body.index("// flux_edge_once: the same three loops")
    tile_body, rest = body[:end], body[end:]
    tile_body = rep(tile_body, "            fluxes[3 * stride + i] = a3; fluxes[4 * stride + i] = a4;\n        }\n        return;\n    }",
                    "            fluxes[3 * stride + i] = a3; fluxes[4 * stride + i] = a4;\n        }\n        PH_MARK(3);\n        return;\n    }")
    k_end = tile_body.rindex("}\n\n")                       # the closing brace of k_flux_tile
    tile_body = tile_body[:k_end] + "    PH_MARK(3);\n" + tile_body[k_end:]
    body = tile_body + rest
    src = src[:j] + body + TAIL
    compile_lib(src)

def run():
    lib_path = os.environ.get("PH_LIB", LIB)
    os.environ["MGCFD_LIB"] = lib_path
    sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
    import bench, mgcfd
    if os.environ.get("PH_MESH") == "tet":                # an unstructured level instead (long rows, unstaged neighbours)
        from mgcfd import meshgen
        mg = meshgen.MultigridMesh(mesh_name="m6wing")
        mg.levels.append(meshgen.make_tet_level(120000, seed=0))
        levels = mgcfd.generated_to_levels(mg)
    else:
        mg, levels = bench.build_workload(67)
    s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    s.set(0, "variables", bench.perturbed_state(s.nel(0), s.far_field()[:5]))
    s.set_option("flux_variant", int(sys.argv[2]) if len(sys.argv) > 2 else 2)
    lib = C.CDLL(lib_path)
    buf = (C.c_ulonglong * (4096 * 8))()
    if os.environ.get("PH_FUSED"):
        import time
        s.smooth(0, 20); s.synchronize()
        lib.mgcfd_debug_phases(buf, 1)
        t0 = time.perf_counter(); s.smooth(0, 66); s.synchronize(); t = (time.perf_counter() - t0) / 198
    else:
        s.bench_flux(0, 20)
        lib.mgcfd_debug_phases(buf, 1)
        t = s.bench_flux(0, 200)
    lib.mgcfd_debug_phases(buf, 1)
    import numpy as np
    a = np.ctypeslib.as_array(buf).reshape(4096, 8).astype(np.float64)
    n = a[:, 7].sum()
    buf = a.sum(0)
    names = ["stage+sync", "edge phase", "sync", "dump+sync", "gather", "boundary+store"]
    if os.environ.get("PH_TILE"):
        names = ["stage+sync", "row loop", "long-row list", "boundary+store", "-", "-"]
    elif os.environ.get("PH_FINE"):
        names = ["entry -> halo ids", "-> q arrived", "derive + LDS store", "sync", "-", "-"]
    print(f"kernel avg {t*1e6:.2f} us, {n} workgroups")
    for k, nm in enumerate(names):
        print(f"  {nm:16s} {buf[k] / n / 100.0:7.3f} us per workgroup")
    print(f"  {'total':16s} {sum(buf[:6]) / n / 100.0:7.3f} us per workgroup")

if __name__ == "__main__":
    (build if sys.argv[1] == "build" else run)()
