#!/usr/bin/env bash
# Build the REAL reference (warwick-hpsc/MG-CFD-app-plain) from its own sources
# where they lie under $MGCFD_REFERENCE (default /root/reference) into
# oracle/_ref/ — test infrastructure used to pin the oracle and to generate
# tests/golden/.  Nothing from the reference is copied into this repository;
# oracle/_ref/ is git-ignored.  The reference's own Makefile writes obj/ and
# bin/ inside its tree (read-only here), so its 13 sources (Makefile:265-277)
# are compiled directly, with the flags SURVEY.md §8c establishes:
#   -fno-fast-math -ffp-contract=off  => flag-independent canonical results.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${MGCFD_REFERENCE:-/root/reference}"
SRC="$REF/src"
OUT="$HERE/_ref"
if [ ! -d "$SRC" ]; then
    echo "build_ref.sh: $SRC not present — skipping reference build" >&2
    exit 0
fi
mkdir -p "$OUT"
FLAGS="-fopenmp -O3 -fno-fast-math -fno-math-errno -ffp-contract=off -w -DPRECISE_FP -DINSN_SET=Host"
INC="-I$SRC -I$SRC/Base -I$SRC/Kernels -I$SRC/Monitoring"
LIBSRC="$SRC/Base/common.cpp $SRC/Base/config.cpp $SRC/Base/io.cpp $SRC/Base/io_enhanced.cpp \
        $SRC/Kernels/flux_loops.cpp $SRC/Kernels/cfd_loops.cpp $SRC/Kernels/mg_loops.cpp \
        $SRC/Kernels/indirect_rw_loop.cpp $SRC/Kernels/validation.cpp \
        $SRC/Monitoring/timer.cpp $SRC/Monitoring/papi_funcs.cpp $SRC/Monitoring/loop_stats.cpp"
# 1) whole binary (timers on, as `make` would with -DTIME)
g++ $FLAGS -DTIME $INC $SRC/euler3d_cpu_double.cpp $LIBSRC -o "$OUT/euler3d_cpu_double_ref.b"
# 1b) the same built with -DLEGACY_ORDERING (edges sorted by (a,b,x,y,z), Makefile:20, src/Base/io.cpp:183-193)
g++ $FLAGS -DTIME -DLEGACY_ORDERING $INC $SRC/euler3d_cpu_double.cpp $LIBSRC -o "$OUT/euler3d_cpu_double_ref_legacy_ordering.b"
# 2) per-kernel harness: reference kernels behind a C ABI (our glue: ref_harness.cpp)
g++ $FLAGS -fPIC -shared $INC "$HERE/ref_harness.cpp" $LIBSRC -o "$OUT/libmgcfd_ref.so"
echo "built $OUT/euler3d_cpu_double_ref.b and $OUT/libmgcfd_ref.so"
