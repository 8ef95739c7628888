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
    # (the export reads the marks of the namespace it is compiled in: PH_NS=fast for the order-free kernel)
    ex_ns = os.environ.get("PH_NS", "exact")
    for ns, contract, extra in (("exact", "off", ["-DMGCFD_PHASE_EXPORT"] if ex_ns == "exact" else []), ("fast", "fast", ["-DMGCFD_PHASE_EXPORT"] if ex_ns == "fast" else [])):
        o = os.path.join(OUT, f"k_phases_{ns}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-Wno-unused-result",
                               "-mllvm", "-amdgpu-kernarg-preload-count=16", f"-ffp-contract={contract}", f"-DMGCFD_KERNEL_NS={ns}", "-DMGCFD_PHASES", *(["-DMGCFD_ORDER_FREE=1"] if ns == "fast" else []), *os.environ.get("PH_DEFS", "").split(),
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
    mg, levels = bench.build_workload(lattice, mesh=os.environ.get("PH_MESH", "lattice"))       # PH_MESH=mixed: the non-uniform-degree level
    s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    s.set(0, "variables", bench.perturbed_state(s.nel(0), s.far_field()[:5]))
    s.set_option("flux_variant", variant)
    if variant & 64: s.set_option("exact", 0)          # the order-free kernel lives in the contracted namespace
    lib = C.CDLL(LIB)
    buf = (C.c_ulonglong * (4096 * 8))()
    if os.environ.get("PH_PROBE"):        # time the indirect_rw probe instead (ablation builds)
        print(f"indirect_rw through the tiles: {s.bench_indirect_rw(0, 200) * 1e6:.2f} us; flux kernel {s.bench_flux(0, 200) * 1e6:.2f} us")
        return
    sweep = os.environ.get("PH_SWEEP") == "1"         # the fused stages of whole sweeps instead of the standalone flux launch
    run_ = (lambda n: (s.smooth(0, n // 3), s.synchronize(), 0.0)[2]) if sweep else (lambda n: s.bench_flux(0, n))
    run_(21)
    lib.mgcfd_debug_phases(buf, 1)
    t = run_(201)
    lib.mgcfd_debug_phases(buf, 1)
    a = np.ctypeslib.as_array(buf).reshape(4096, 8).astype(np.float64)
    n = a[:, 7].sum()
    tot = a.sum(0)
    if not (variant & (64 | 32)):
        names = ["loads + derive + stage (thread 0)", "staging barrier", "row pairs", "boundary rows + epilogue / store", "-", "-"]
    elif variant & 64:
        names = ["loads + derive + stage (thread 0)", "staging barrier", "half rows + LDS adds", "barrier", "own sums + boundary + store", "-"]
    else:
        names = ["stage + barrier", "half rows", "barrier (records dead)", "hand-over + barrier", "ordered adds", "boundary + store"]
    print(f"variant {variant} lattice {lattice}: kernel avg {t*1e6:.2f} us, {n:.0f} workgroups timed")
    for k, nm in enumerate(names):
        print(f"  {nm:24s} {tot[k] / n / 100.0:7.3f} us per workgroup")
    print(f"  {'total':24s} {sum(tot[:6]) / n / 100.0:7.3f} us per workgroup")
    for lo, hi, nm in ((0, 768, "workgroups 0-767 (first on their CU slot)"), (768, 4096, "workgroups 768- (behind another)")):
        nn = a[lo:hi, 7].sum()
        if nn > 0: print(f"  {nm}: " + " | ".join(f"{a[lo:hi, k].sum() / nn / 100.0:.2f}" for k in range(6)))
    if hasattr(lib, "mgcfd_debug_phase_abs"):
        # the LAST launch: absolute times of every mark (100 MHz ticks) and where each workgroup ran
        ab = (C.c_ulonglong * (4096 * 8))()
        lib.mgcfd_debug_phase_abs(ab)
        nb = min(4096, s.nel(0) // 256 + (1 if s.nel(0) % 256 else 0))
        b = np.ctypeslib.as_array(ab).reshape(4096, 8)[:nb].astype(np.int64)
        nmark = max(k for k in range(6) if b[:, 1 + k].max() > 0) + 1
        t0 = b[:, 0].min()
        st, en = (b[:, 0] - t0) / 100.0, (b[:, nmark] - t0) / 100.0
        print(f"  last launch: workgroups begin 0 .. {st.max():.2f} us, end {en.min():.2f} .. {en.max():.2f} us; lifetime mean {np.mean(en - st):.2f} max {np.max(en - st):.2f}")
        hist, edges = np.histogram(st, bins=12)
        print("  begin-time histogram:", " ".join(f"{e:.1f}:{h}" for h, e in zip(hist, edges[:-1])))
        hist, edges = np.histogram(en, bins=12)
        print("  end-time histogram:  ", " ".join(f"{e:.1f}:{h}" for h, e in zip(hist, edges[:-1])))
        first = st < 1.0
        for nm, sel in (("first round (begin < 1 us)", first), ("later rounds", ~first)):
            if sel.sum() == 0: continue
            print(f"  {nm}: {sel.sum()} workgroups; time of each mark since the launch's first workgroup began, us (min / 10% / median / 90% / max)")
            for k in range(nmark + 1):
                v = (b[sel, k] - t0) / 100.0
                q = np.percentile(v, [0, 10, 50, 90, 100])
                print(f"    {'begin' if k == 0 else 'mark %d' % (k - 1):8s} " + " / ".join(f"{x:6.2f}" for x in q))
        # end times by dispatch order (blockIdx): do the workgroups dispatched first also finish first?
        for lo, hi in ((0, 151), (151, 512), (512, 1024), (1024, nb)):
            if hi > lo and lo < nb:
                e = en[lo:min(hi, nb)]
                print(f"  workgroups {lo}-{min(hi, nb) - 1}: end min / median / max {e.min():.2f} / {np.median(e):.2f} / {e.max():.2f} us")
        hw = b[:, 7] & 0xFFFFFFFF; xcc = (b[:, 7] >> 32) & 0xF
        cu = (xcc << 16) | (hw & 0xFF00)                      # XCC_ID, SE_ID/SH_ID/CU_ID of HW_ID
        ids, cnt = np.unique(cu, return_counts=True)
        last_end = np.array([en[cu == c].max() for c in ids])
        print(f"  {len(ids)} CUs ran the launch; workgroups per CU: " + " ".join(f"{k}:{(cnt == k).sum()}" for k in sorted(set(cnt))))
        for k in sorted(set(cnt)):
            print(f"    CUs with {k} workgroups: last end mean {last_end[cnt == k].mean():.2f} max {last_end[cnt == k].max():.2f} us")
        per_x = [f"{x}:{(xcc == x).sum()}/{en[xcc == x].max():.1f}" for x in sorted(set(xcc))]
        print("  per XCD workgroups / last end:", " ".join(per_x))

if __name__ == "__main__":
    (build if sys.argv[1] == "build" else run)()
