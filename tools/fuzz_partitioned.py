#!/usr/bin/env python3
"""Randomised partitioned-level / partitioned-hierarchy sweep (GPU box): a random level (lattice, tetrahedra, hub, random graph) split into
2-5 parts by a random partitioner, every part a solver of its own on this GPU with ghost nodes and halo exchanges after
every Runge-Kutta stage (tests/test_gpu_parity.py: _partitioned_level_check), against the unpartitioned run, bit for bit.
    python tools/fuzz_partitioned.py [--seeds 40] [--first 0]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=40)
    ap.add_argument("--first", type=int, default=0)
    args = ap.parse_args()
    import numpy as np
    from mgcfd import meshgen
    import fuzz_parity
    import test_gpu_parity as T
    bad = 0
    for seed in range(args.first, args.first + args.seeds):
        rng = np.random.default_rng(5000 + seed)
        while True:
            kind, name, mg, _ = fuzz_parity.make_case(rng)
            if 20 <= mg.levels[0].nel <= 5000:
                break
        fvcorr = name == "fvcorr"                           # (local time step: the Python helpers assume the global one; the library's group loop takes both)
        # the whole hierarchy too (multigrid cycles over the partitions) where the generator made one with coarser levels smaller
        hier = None
        if len(mg.levels) > 1 and all(mg.levels[k + 1].nel < mg.levels[k].nel for k in range(len(mg.levels) - 1)) and mg.levels[-1].nel >= 8:
            import copy
            hier = copy.deepcopy(mg)
        mg.levels = mg.levels[:1]
        mg.levels[0].mg_map = None
        n_parts = int(rng.integers(2, 6))
        partitioner = str(rng.choice(["slab", "rcb"]))
        fused = bool(rng.integers(2))
        variant = int(rng.choice([-1, 0, 1, 2, 3, 16, 32]))
        tag = f"seed {seed}: {kind} {name} {mg.levels[0].nel} nodes, {n_parts} parts ({partitioner}), fused={fused}, variant={variant}"
        try:
            if hier is not None and not fvcorr:
                T._partitioned_hierarchy_check(hier, n_parts, fused, cycles=int(rng.integers(1, 4)))
            if not fvcorr:
                T._partitioned_level_check(mg, variant, n_parts, partitioner, fused, sweeps=int(rng.integers(1, 4)), seed=seed)
            # ... and the same level by the library's own loop over an in-process group (mgcfd_group_sweeps)
            import test_gpu_configs as TC
            TC._group_sweeps_check(mg, n_parts, int(rng.integers(1, 7)), partitioner=partitioner)     # (4 and more: a host thread per rank)
            print(tag + (f" + hierarchy {[l.nel for l in hier.levels]}" if hier is not None else "") + ": ok", flush=True)
        except AssertionError as e:
            if os.environ.get("FUZZ_RAISE"): raise                  # (the traceback says which comparison it was)
            print(tag + ": MISMATCH " + str(e)[:200], flush=True)
            bad += 1
        except Exception as e:                              # (a random mesh whose state goes invalid: the reference would exit too)
            if "ERR_NAN" in str(e) or "ERR_NEG" in str(e):
                print(tag + ": the state goes invalid, skipped", flush=True)
            else:
                raise
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
