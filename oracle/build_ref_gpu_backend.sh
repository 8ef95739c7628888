#!/usr/bin/env bash
# Face 2 of the drop-in boundary, compiled: the REFERENCE's own main() (src/euler3d_cpu_double.cpp) with its own
# src/Base/*, src/Monitoring/* and validation.cpp, where they lie under $MGCFD_REFERENCE (default /root/reference), and
# mg-cfd-app-plain_amd/binding/gpu_backend.cpp IN PLACE OF flux_loops.cpp, cfd_loops.cpp, mg_loops.cpp and
# indirect_rw_loop.cpp, linked against libmgcfd_hip.so.  Output: oracle/_ref/euler3d_ref_main_gpu_backend.b (git-ignored;
# it travels to the GPU box, the reference sources do not).  tests/test_gpu_binding.py runs it on golden inputs and
# compares the variables dump with the reference binary's, byte for byte.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REPO="$(dirname "$HERE")"
REF="${MGCFD_REFERENCE:-/root/reference}"
SRC="$REF/src"
OUT="$HERE/_ref"
if [ ! -d "$SRC" ]; then
    echo "build_ref_gpu_backend.sh: $SRC not present — skipping" >&2
    exit 0
fi
LIBDIR="$REPO/mg-cfd-app-plain_amd/csrc"
if [ ! -f "$LIBDIR/libmgcfd_hip.so" ]; then
    echo "build_ref_gpu_backend.sh: build $LIBDIR/libmgcfd_hip.so first (make -C $LIBDIR)" >&2
    exit 1
fi
mkdir -p "$OUT"
FLAGS="-fopenmp -O3 -fno-fast-math -fno-math-errno -ffp-contract=off -w -DTIME -DPRECISE_FP -DINSN_SET=Host"
INC="-I$SRC -I$SRC/Base -I$SRC/Kernels -I$SRC/Monitoring -I$REPO/include"
KEPT="$SRC/Base/common.cpp $SRC/Base/config.cpp $SRC/Base/io.cpp $SRC/Base/io_enhanced.cpp $SRC/Kernels/validation.cpp \
      $SRC/Monitoring/timer.cpp $SRC/Monitoring/papi_funcs.cpp $SRC/Monitoring/loop_stats.cpp"
g++ $FLAGS $INC "$SRC/euler3d_cpu_double.cpp" $KEPT "$REPO/mg-cfd-app-plain_amd/binding/gpu_backend.cpp" \
    -L"$LIBDIR" -lmgcfd_hip -Wl,-rpath,'$ORIGIN/../../mg-cfd-app-plain_amd/csrc' -Wl,-rpath-link,/opt/rocm/lib \
    -o "$OUT/euler3d_ref_main_gpu_backend.b"
echo "built $OUT/euler3d_ref_main_gpu_backend.b"
