#!/usr/bin/env python3
"""Randomised runs of the drop-in binary (GPU box): a random hierarchy written in the reference's file formats,
euler3d_gpu_double with random flags (-g, -m, --no-timers, --legacy-ordering, --no-indirect-rw, --gpus N), its variables dump,
RMS lines and LoopNumIters counts against the oracle reading the same files.
    python tools/fuzz_driver.py [--seeds 40] [--first 0]"""
import argparse, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
EXE = os.path.join(ROOT, "mg-cfd-app-plain_amd", "csrc", "euler3d_gpu_double")
# the REFERENCE's own main() linked against libmgcfd_hip.so through the reference-side binding (oracle/build_ref_gpu_backend.sh;
# built in the build container, travels with the repository): run on the same input where it exists
EXE_BINDING = os.path.join(ROOT, "oracle", "_ref", "euler3d_ref_main_gpu_backend.b")


def same_bits_or_both_nan(np, got, want):
    """Bit for bit — except that a NaN equals a NaN.  A run can END with a NaN the reference never reports (it checks after a
    time_step only, and the last operation of a cycle is a prolongation: seed 10105); an x86 core then holds the negative
    default NaN where a GPU holds the positive one ("-nan" against "nan" in the dump), which no arithmetic rule fixes."""
    got, want = np.ascontiguousarray(got), np.ascontiguousarray(want)
    return got.shape == want.shape and bool(np.all((got.view(np.int64) == want.view(np.int64)) | (np.isnan(got) & np.isnan(want))))


def _one(seed, np, meshgen, oracle, fuzz_parity, read_loop_iters):
    bad = 0
    rng = np.random.default_rng(9000 + seed)
    gpu_rng = np.random.default_rng(19000 + seed)           # (its own stream: the cases of the seeds stay as they were)
    while True:
        kind, name, mg, _ = fuzz_parity.make_case(rng)
        # the file format lists a level's coarser neighbour by index: keep hierarchies the reference itself accepts
        if max(l.nel for l in mg.levels) <= 3000 and all(mg.levels[k + 1].nel <= mg.levels[k].nel for k in range(len(mg.levels) - 1)):
            break
    d = tempfile.mkdtemp(prefix="mgcfd_fuzz_")
    try:
        meshgen.write_input(mg, d)
        cycles = int(rng.integers(1, 5))
        dup = int(rng.choice([1, 1, 2, 3]))
        legacy = bool(rng.integers(2)) and dup == 1
        flags = [f for f, on in (("--no-timers", rng.integers(2)), ("--no-indirect-rw", rng.integers(2))) if on]
        # several ranks (all on this one GPU): a single level is partitioned over them (halo stores after every stage, a host
        # thread per rank), a hierarchy runs one level per rank
        gpus = int(gpu_rng.choice([1, 1, 2, 3, 4]))
        if gpus > 1: flags += ["--gpus", str(gpus), "--gpus-share-device"]
        cmd = [EXE, "-i", "input.dat", "-d", d, "-o", d + "/", "-g", str(cycles), "-m", str(dup), "--output-variables"] + flags + (["--legacy-ordering"] if legacy else [])
        tag = f"seed {seed}: {kind} {name} {[l.nel for l in mg.levels]} -g {cycles} -m {dup} {' '.join(flags)}{' --legacy-ordering' if legacy else ''}"
        oc = oracle.OracleCase.from_input_dat(os.path.join(d, "input.dat"), dup, legacy_ordering=legacy)
        rc, rms, iters = oc.solve(cycles, run_indirect_rw=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if rc != 0:
            ok = r.returncode == 1 and "ERROR:" in r.stdout
            print(tag + (": both abort: ok" if ok else f": MISMATCH (oracle aborts with {rc}, driver rc {r.returncode})"), flush=True)
            bad += 0 if ok else 1
            return bad
        problems = []
        if r.returncode != 0:
            problems.append(f"driver rc {r.returncode}: {r.stdout[-200:]} | {r.stderr[-400:]}")
        else:
            got = np.loadtxt(os.path.join(d, f"variables.size={dup}x.cycles={cycles}.level=0")).reshape(-1, 5)
            want = oc.array(0, "variables").reshape(-1, 5)
            if not same_bits_or_both_nan(np, got, want):
                problems.append("variables dump differs")
            lines = [l for l in r.stdout.splitlines() if "RMS" in l]
            if len(lines) != cycles or any(f"(RMS = {rms[c]:.3e})" not in lines[c] for c in range(cycles)):
                problems.append("RMS lines differ")
            got_it = read_loop_iters(os.path.join(d, "LoopNumIters.csv"), oc.nlevels)
            for l in range(oc.nlevels):
                want_it = {"flux": iters[l].flux, "update": 0, "compute_step": iters[l].compute_step, "time_step": iters[l].time_step,
                           "restrict": iters[l].restrict_, "prolong": iters[l].prolong}
                if {k: got_it[l][k] for k in want_it} != want_it:
                    problems.append(f"LoopNumIters level {l}: {got_it[l]} vs {want_it}")
        # (a single-level input that is not fvcorr: the reference never reads its .coords and aborts in adjust_ewt, SURVEY.md §7 —
        #  the library and the oracle read them; nothing to compare there)
        if os.path.exists(EXE_BINDING) and not legacy and not problems and (len(mg.levels) > 1 or name == "fvcorr"):
            d2 = tempfile.mkdtemp(prefix="mgcfd_fuzz_b_")
            try:
                rb = subprocess.run([EXE_BINDING, "-i", "input.dat", "-d", d, "-o", d2 + "/", "-g", str(cycles), "-m", str(dup), "--output-variables"],
                                    capture_output=True, text=True, cwd=d2, env=dict(os.environ, OMP_NUM_THREADS="1"))
                if rb.returncode != 0:
                    problems.append(f"reference main on the library: rc {rb.returncode}: {rb.stdout[-200:]}")
                else:
                    got_b = np.loadtxt(os.path.join(d2, f"variables.size={dup}x.cycles={cycles}.level=0")).reshape(-1, 5)
                    if not same_bits_or_both_nan(np, got_b, want):
                        problems.append("reference main on the library: variables dump differs")
                    lines_b = [l for l in rb.stdout.splitlines() if "RMS" in l]
                    if len(lines_b) != cycles or any(f"(RMS = {rms[c]:.3e})" not in lines_b[c] for c in range(cycles)):
                        problems.append("reference main on the library: RMS lines differ")
                tag += " + reference main on the library"
            finally:
                shutil.rmtree(d2, ignore_errors=True)
        print(tag + (": ok" if not problems else ": MISMATCH " + "; ".join(problems)), flush=True)
        bad += 1 if problems else 0
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return bad


def run_seeds(first, count):
    """Returns the number of mismatching seeds (prints one line per seed)."""
    import numpy as np
    from mgcfd import meshgen
    import oracle_py as oracle
    import fuzz_parity
    from test_oracle_golden import read_loop_iters
    bad = 0
    for seed in range(first, first + count):
        bad += _one(seed, np, meshgen, oracle, fuzz_parity, read_loop_iters)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=40)
    ap.add_argument("--first", type=int, default=0)
    args = ap.parse_args()
    bad = run_seeds(args.first, args.seeds)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
