#!/usr/bin/env python3
"""Build a library variant from a patched copy of csrc/kernels.hip:
   tools/exp_build.py NAME 'old text' 'new text' ['old2' 'new2' ...]   ->  csrc/build/exp/libmgcfd_hip_NAME.so
(timing experiments only; the host objects of the current build are reused)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mg-cfd-app-plain_amd", "csrc")
OUT = os.path.join(CSRC, "build", "exp")
name, pairs = sys.argv[1], sys.argv[2:]
src = open(os.path.join(CSRC, "kernels.hip")).read()
for a, b in zip(pairs[0::2], pairs[1::2]):
    a = a.encode().decode("unicode_escape"); b = b.encode().decode("unicode_escape")
    assert src.count(a) >= 1, a
    src = src.replace(a, b)
os.makedirs(OUT, exist_ok=True)
f = os.path.join(OUT, f"kernels_{name}.hip")
open(f, "w").write(src)
objs = []
for ns, contract in (("exact", "off"), ("fast", "fast")):
    o = os.path.join(OUT, f"k_{name}_{ns}.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-Wno-unused-result",
                           "-mllvm", "-amdgpu-kernarg-preload-count=16",      # (as the Makefile's DEVFLAGS: without it every launch reads 0.5 us slower)
                           f"-ffp-contract={contract}", f"-DMGCFD_KERNEL_NS={ns}"] + (["-DMGCFD_ORDER_FREE=1"] if ns == "fast" else []) + [ f"-I{ROOT}/include", f"-I{CSRC}", "-c", f, "-o", o])
    objs.append(o)
host = [os.path.join(CSRC, "build", x) for x in ("solver.o", "mesh.o", "preprocess.o")]
lib = os.path.join(OUT, f"libmgcfd_hip_{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + host)
print("built", lib)
