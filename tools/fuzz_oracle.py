#!/usr/bin/env python3
"""Randomised pinning of the ORACLE (build container: needs oracle/_ref): random hierarchies written in the reference's
file formats, the REAL reference binary (-g, -m; its -DLEGACY_ORDERING build for some) against oracle/mgcfd_oracle.c
reading the same files — variables dump bit for bit, RMS lines, LoopNumIters counts, and for runs the reference aborts
the same error class and cell.
    python tools/fuzz_oracle.py [--seeds 100] [--first 0]"""
import argparse, ctypes as C, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=100)
    ap.add_argument("--first", type=int, default=0)
    args = ap.parse_args()
    import numpy as np
    from mgcfd import meshgen
    import oracle_py as O
    import fuzz_parity
    from test_oracle_golden import read_loop_iters
    if not O.have_reference():
        raise SystemExit("oracle/_ref is missing: run oracle/build_ref.sh first")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    bad = 0
    for seed in range(args.first, args.first + args.seeds):
        rng = np.random.default_rng(11000 + seed)
        while True:
            kind, name, mg, _ = fuzz_parity.make_case(rng)
            if max(l.nel for l in mg.levels) <= 3000 and all(mg.levels[k + 1].nel <= mg.levels[k].nel for k in range(len(mg.levels) - 1)):
                break
        d = tempfile.mkdtemp(prefix="mgcfd_or_")
        try:
            meshgen.write_input(mg, d)
            cycles = int(rng.integers(1, 6))
            dup = int(rng.choice([1, 1, 2, 3]))
            legacy = bool(rng.integers(2)) and dup == 1
            r = subprocess.run([O.REF_BIN_LEGACY if legacy else O.REF_BIN, "-i", "input.dat", "-d", d, "-o", d + "/", "-g", str(cycles), "-m", str(dup),
                                "--output-variables"], capture_output=True, text=True, env=env)
            oc = O.OracleCase.from_input_dat(os.path.join(d, "input.dat"), dup, legacy_ordering=legacy, coords_as_reference=True)
            rc, rms, iters = oc.solve(cycles, run_indirect_rw=True)
            tag = f"seed {seed}: {kind} {name} {[l.nel for l in mg.levels]} -g {cycles} -m {dup}{' legacy-ordering' if legacy else ''}"
            problems = []
            if r.returncode != 0:
                err = next((l for l in r.stdout.splitlines() if l.startswith("ERROR")), "")
                cell_line = next((l for l in r.stdout.splitlines() if l.startswith("Cell")), "")
                want_err = {1: "ERROR: NaN detected!", 2: "ERROR: Negative density detected!", 3: "ERROR: Negative density.energy detected!"}.get(rc)
                if rc == 0 or err != want_err:
                    problems.append(f"reference aborted ({err!r}, rc {r.returncode}) but the oracle returned {rc}")
                else:
                    # which level failed is not in the reference's message: the cell must be the first bad one of SOME level
                    cell = int(cell_line.split(":")[0].split()[1])
                    cells = []
                    for l in range(oc.nlevels):
                        b = C.c_int64(-1)
                        if oc.lib.ora_check_for_invalid_variables(oc.levels[l].variables, oc.levels[l].nel, C.byref(b)) == rc:
                            cells.append(b.value)
                    if cell not in cells:
                        problems.append(f"reference names cell {cell}, oracle's first bad cells {cells}")
            elif rc != 0:
                problems.append(f"oracle aborted with {rc}, the reference did not")
            else:
                got = oc.array(0, "variables").reshape(-1, 5)
                want = np.loadtxt(os.path.join(d, f"variables.size={dup}x.cycles={cycles}.level=0")).reshape(-1, 5)
                if not np.array_equal(np.ascontiguousarray(got).view(np.int64), want.view(np.int64)):
                    problems.append("variables dump differs")
                lines = [l for l in r.stdout.splitlines() if "RMS" in l]
                if len(lines) != cycles or any(f"(RMS = {rms[c]:.3e})" not in lines[c] for c in range(cycles)):
                    problems.append("RMS lines differ")
                want_it = read_loop_iters(os.path.join(d, "LoopNumIters.csv"), oc.nlevels)
                for l in range(oc.nlevels):
                    g = {"flux": iters[l].flux, "update": 0, "compute_step": iters[l].compute_step, "time_step": iters[l].time_step,
                         "restrict": iters[l].restrict_, "prolong": iters[l].prolong, "indirect_rw": iters[l].indirect_rw}
                    if g != want_it[l]:
                        problems.append(f"LoopNumIters level {l}: {g} vs {want_it[l]}")
            oc.close()
            print(tag + (": ok" if not problems else ": MISMATCH " + "; ".join(problems)), flush=True)
            bad += 1 if problems else 0
        finally:
            shutil.rmtree(d, ignore_errors=True)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
