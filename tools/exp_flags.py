#!/usr/bin/env python3
"""Build a library variant of the CURRENT csrc/kernels.hip with extra compiler flags (the -DMGCFD_EXP_* experiment switches):
   tools/exp_flags.py NAME -DMGCFD_EXP_X[=v] ...   ->  csrc/build/exp/libmgcfd_hip_NAME.so
(timing experiments only; the host objects of the current build are reused; select with MGCFD_LIB=...)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mg-cfd-app-plain_amd", "csrc")
OUT = os.path.join(CSRC, "build", "exp")
name, flags = sys.argv[1], sys.argv[2:]
src = os.environ.get("MGCFD_EXP_SRC") or os.path.join(CSRC, "kernels.hip")
os.makedirs(OUT, exist_ok=True)
procs, objs = [], []
for ns, contract in (("exact", "off"), ("fast", "fast")):
    o = os.path.join(OUT, f"k_{name}_{ns}.o")
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-Wno-unused-result",
                                   "-mllvm", "-amdgpu-kernarg-preload-count=" + os.environ.get("MGCFD_EXP_PRELOAD", "16"), f"-ffp-contract={contract}", f"-DMGCFD_KERNEL_NS={ns}"]
                                  + (["-DMGCFD_ORDER_FREE=1"] if ns == "fast" else []) + flags
                                  + [f"-I{ROOT}/include", f"-I{CSRC}", "-c", src, "-o", o]))
    objs.append(o)
for p in procs:
    if p.wait() != 0: sys.exit(1)
host = [os.path.join(CSRC, "build", x) for x in ("solver.o", "mesh.o", "preprocess.o")]
lib = os.path.join(OUT, f"libmgcfd_hip_{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + host)
print("built", lib)
