#!/usr/bin/env python3
"""One seed of tools/fuzz_parity.py --fast in detail: how far the fast mode's state and RMS history are from the oracle's.
   python tools/exp/fuzz_one.py SEED [repeats=3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import mgcfd, oracle_py as oracle, test_gpu_parity as T, fuzz_parity as F
seed = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(seed)
kind, name, mg, cycles = F.make_case(rng)
levels = mgcfd.generated_to_levels(mg)
want, want_rms = T._oracle_solve_arrays(oracle, levels, mg.mesh_variant, cycles)
opts = {"fuse_update": int(rng.integers(0, 2)), "flux_variant": int(rng.choice([-1, 0, 1, 2, 3, 16, 32, 33])), "graph": int(rng.integers(0, 2))}
print(kind, name, [l.nel for l in mg.levels], opts, "cycles", cycles)
lib = oracle.load()
for r in range(reps):
    s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    for k, v in opts.items(): s.set_option(k, v)
    s.set_option("exact", 0)
    rms = s.run_cycles(cycles)
    out = []
    for l in range(len(levels)):
        got = s.get(l, "variables")
        rel = np.max(np.abs(got - want[l]) / np.maximum(np.abs(want[l]), 1e-300))
        out.append((l, float(rel), int(lib.ora_identify_differences(oracle.ptr(np.ascontiguousarray(got)), oracle.ptr(np.ascontiguousarray(want[l])), levels[l]["nel"], mg.mesh_variant))))
    print(f"run {r}: per level (level, max relative difference, first value out of the -v tolerance or -1): {out}; rms rel diff {np.max(np.abs(np.asarray(rms) - want_rms) / np.abs(want_rms)):.3e}; rms {rms}")
    s.close()
