"""Face 2 of the drop-in boundary, exercised: the REFERENCE's own main() (src/euler3d_cpu_double.cpp, compiled where it lies
by oracle/build_ref_gpu_backend.sh) linked against libmgcfd_hip.so through the reference-side binding
mg-cfd-app-plain_amd/binding/gpu_backend.cpp — forwarding functions with the reference's own signatures
(src/Kernels/flux_loops.h:30-43, cfd_loops.h:13-42, mg_loops.h:27-33) in place of its three *_loops.cpp files.
The binary is built in the build container and travels with the repository (oracle/_ref/, git-ignored)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "oracle", "_ref", "euler3d_ref_main_gpu_backend.b")


def _case(name):
    d = os.path.join(GOLDEN, name)
    meta = dict(l.strip().split(" = ") for l in open(os.path.join(d, "case.txt")))
    return d, int(meta["cycles"]), int(meta["duplicate"])


def _csv_row(path):
    rows = [l.rstrip(",\n").split(",") for l in open(path) if l.strip()]
    return dict(zip(rows[0], rows[1]))


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/euler3d_ref_main_gpu_backend.b not built (needs the reference sources: oracle/build_ref_gpu_backend.sh)")
@pytest.mark.parametrize("case", ["m6_3lvl", "m6_2lvl_dup2", "fvcorr_1lvl"])
def test_reference_main_runs_on_the_gpu_library(case, tmp_path):
    """The reference's main() drives its cycle loop; every flux / step-factor / time_step / indirect_rw / restrict /
    prolong call lands in libmgcfd_hip.so.  The variables dump must be the reference binary's byte for byte, the RMS
    lines and the loop counters the same."""
    d, cycles, dup = _case(case)
    cmd = [EXE, "-i", "input.dat", "-d", os.path.join(d, "input"), "-o", str(tmp_path) + "/", "-g", str(cycles), "-m", str(dup), "--output-variables"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path), env=dict(os.environ, OMP_NUM_THREADS="1"), timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    dump = tmp_path / f"variables.size={dup}x.cycles={cycles}.level=0"
    assert dump.read_bytes() == open(os.path.join(d, "variables.level0.txt"), "rb").read()
    want_lines = [l.strip() for l in open(os.path.join(d, "stdout.txt")) if "RMS" in l]
    got_lines = [l.strip() for l in r.stdout.splitlines() if "RMS" in l]
    assert got_lines == want_lines
    want, got = _csv_row(os.path.join(d, "LoopNumIters.csv")), _csv_row(tmp_path / "LoopNumIters.csv")
    for k in want:
        if k[:-1] in ("flux", "update", "compute_step", "time_step", "restrict", "prolong", "indirect_rw"):
            assert got[k] == want[k], k
    # the library really was in the process (the binding has no host fallback: without it the link would not resolve)
    ldd = subprocess.run(["ldd", EXE], capture_output=True, text=True).stdout
    assert "libmgcfd_hip.so" in ldd
