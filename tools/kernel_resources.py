#!/usr/bin/env python3
"""Registers, scratch (spills), LDS and occupancy of every kernel instantiation, from hipcc's own remarks
(-Rpass-analysis=kernel-resource-usage) on csrc/kernels.hip; build container, no GPU.   python tools/kernel_resources.py [filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "mg-cfd-app-plain_amd", "csrc")
flt = sys.argv[1] if len(sys.argv) > 1 else ""
NS = os.environ.get("NS", "exact")             # NS=fast: the contracted namespace (with its order-free kernels)
NSFLAGS = ["-ffp-contract=fast", "-DMGCFD_KERNEL_NS=fast", "-DMGCFD_ORDER_FREE=1"] if NS == "fast" else ["-ffp-contract=off", "-DMGCFD_KERNEL_NS=exact"]
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-Wno-unused-result",
                    "-mllvm", "-amdgpu-kernarg-preload-count=16"] + NSFLAGS + [f"-I{ROOT}/include", f"-I{CS}",
                    "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CS, "kernels.hip"), "-o", "/tmp/kernel_resources.o"],
                   capture_output=True, text=True)
blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)
names = [b.split("\n")[0].strip() for b in blocks[1:]]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
print("VGPR SGPR scratch occ    LDS  kernel")
for b, d in zip(blocks[1:], dem):
    g = lambda k: int(m.group(1)) if (m := re.search(k + r": (\d+)", b)) else -1
    if flt in d:
        sc, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
        print(f"{g('VGPRs'):4d} {g('SGPRs'):4d} {sc:7d} {occ:3d} {lds:6d}  {d[:150]}")
