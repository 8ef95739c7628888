#!/usr/bin/env bash
# Device assembly of csrc/kernels.hip for one namespace: tools/kernel_isa.sh fast|exact OUT.s [-DMGCFD_EXP_...]
set -e
ns=$1; out=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd); CS=$ROOT/mg-cfd-app-plain_amd/csrc
if [ "$ns" = fast ]; then nsf="-ffp-contract=fast -DMGCFD_KERNEL_NS=fast -DMGCFD_ORDER_FREE=1"; else nsf="-ffp-contract=off -DMGCFD_KERNEL_NS=exact"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-device-only -O3 -std=c++17 -fno-fast-math -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=16 \
  $nsf "$@" -I$ROOT/include -I$CS -S ${MGCFD_EXP_SRC:-$CS/kernels.hip} -o $out
