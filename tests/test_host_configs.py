"""CPU-side tests of BASELINE.json's configurations and of bench.py's launcher.

* configs[0] at size: the ORACLE against the REFERENCE BINARY on the fvcorr-like 97,335-node input, 1,000 iterations
  (needs oracle/_ref, i.e. the build container; the GPU test test_cfg1_* then compares the HIP path with the oracle
  on the same input).
* bench.py --gpus N from a plain invocation starts N ranks itself; N must equal WORLD_SIZE under a launcher.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "euler3d_cpu_double_ref.b")


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref not built (needs the reference sources)")
def test_cfg1_oracle_equals_reference_binary_at_97k_nodes_1000_iterations(oracle, tmp_path):
    """The reference binary (oracle/_ref, -g 1000 --output-variables) and the oracle read the same files: the dumped
    variables bit for bit, the RMS lines, the loop counters."""
    from mgcfd import meshgen
    from test_oracle_golden import read_loop_iters
    d = str(tmp_path)
    mg = meshgen.make_multigrid((46,), "fvcorr", seed=0, cavity_radius=0.001)
    assert mg.levels[0].nel == 97335
    meshgen.write_input(mg, d)
    iters = 1000
    r = subprocess.run([REF_BIN, "-i", "input.dat", "-d", d, "-o", d + "/", "-g", str(iters), "--output-variables"],
                       capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS="1"), cwd=d, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]
    want = np.loadtxt(os.path.join(d, f"variables.size=1x.cycles={iters}.level=0")).reshape(-1, 5)
    oc = oracle.OracleCase.from_input_dat(os.path.join(d, "input.dat"), coords_as_reference=True)
    rc, rms, it = oc.solve(iters, run_indirect_rw=True)
    assert rc == 0
    got = np.ascontiguousarray(oc.array(0, "variables").reshape(-1, 5))
    assert np.array_equal(got.view(np.int64), want.view(np.int64)), f"max abs diff {np.abs(got - want).max():.3e}"
    lines = [ln for ln in r.stdout.splitlines() if "RMS" in ln]
    assert len(lines) == iters and all(f"(RMS = {rms[c]:.3e})" in lines[c] for c in range(iters))
    want_it = read_loop_iters(os.path.join(d, "LoopNumIters.csv"), 1)[0]
    assert want_it["flux"] == it[0].flux and want_it["time_step"] == it[0].time_step and want_it["indirect_rw"] == it[0].indirect_rw
    assert rms[-1] < 0.2 * rms[0]                                   # a developed, converging flow
    oc.close()


def _bench(args, env=None, timeout=600):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=timeout)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: two child ranks rendezvous (gloo, no GPU: --plumbing-only), build a
    2-way partition that covers the level, and rank 0 reports n_gpus = 2 from the process group."""
    r = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--plumbing-only"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["plumbing_only"] and line["config"]["workload_kind"] == "partitioned"
    assert line["config"]["halo_nodes_total"] > 0


def test_bench_refuses_a_world_size_other_than_gpus():
    r = _bench(["--gpus", "4", "--plumbing-only"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_refuses_more_ranks_than_gpus():
    """Without the rehearsal switch a box with fewer GPUs than --gpus must fail loudly, never run fewer ranks."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    r = _bench(["--gpus", "2"])
    assert r.returncode == 2 and "refusing" in r.stderr
