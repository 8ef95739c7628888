#!/usr/bin/env python3
"""Randomised file-boundary sweep (build container: needs oracle/_ref): random hierarchies written in the reference's
formats, read back by (a) the REFERENCE's read_grid through oracle/ref_harness.cpp, (b) the oracle's reader, (c) the
library's reader (mgcfd_mesh_load) — sizes, edge arrays (class order, signs), volumes and coordinates must be identical.
    python tools/fuzz_reader.py [--seeds 60] [--first 0]"""
import argparse, os, shutil, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=60)
    ap.add_argument("--first", type=int, default=0)
    args = ap.parse_args()
    import numpy as np
    import mgcfd
    from mgcfd import meshgen
    import oracle_py as O
    import fuzz_parity
    if not O.have_reference():
        raise SystemExit("oracle/_ref is missing: run oracle/build_ref.sh first")
    ref = O.load_reference()
    bad = 0
    for seed in range(args.first, args.first + args.seeds):
        rng = np.random.default_rng(7000 + seed)
        while True:
            kind, name, mg, _ = fuzz_parity.make_case(rng)
            if max(l.nel for l in mg.levels) <= 3000 and all(mg.levels[k + 1].nel <= mg.levels[k].nel for k in range(len(mg.levels) - 1)):
                break
        d = tempfile.mkdtemp(prefix="mgcfd_rd_")
        try:
            meshgen.write_input(mg, d)
            info = O.parse_input_dat(os.path.join(d, "input.dat"))
            n_levels = len(info["levels"])
            ref.ref_init(n_levels, info["mesh_variant"])
            ours = mgcfd.Mesh("input.dat", d)
            oc = O.OracleCase.from_input_dat(os.path.join(d, "input.dat"))
            problems = []
            for l in range(n_levels):
                sizes = np.zeros(8, dtype=np.int64)
                ref.ref_set_level(l)
                ref.ref_read_grid(os.path.join(d, info["levels"][l]).encode(), O.ptr(sizes))
                nel, n_edges = int(sizes[0]), int(sizes[1])
                vol, edges, coords = np.zeros(nel), np.zeros(n_edges, dtype=O.EDGE_DTYPE), np.zeros((nel, 3))
                ref.ref_grid_copy(O.ptr(vol), O.ptr(edges), O.ptr(coords))
                L = ours.level(l)
                if (L["nel"], L["n_internal"], L["n_boundary"], L["n_wall"]) != (nel, int(sizes[2]), int(sizes[3]), int(sizes[4])):
                    problems.append(f"level {l} sizes {(L['nel'], L['n_internal'], L['n_boundary'], L['n_wall'])} vs {sizes[:5].tolist()}")
                    continue
                if not np.array_equal(L["edges"], edges): problems.append(f"level {l} edges (library)")
                if not np.array_equal(oc.edges(l), edges): problems.append(f"level {l} edges (oracle)")
                if not np.array_equal(L["volumes"], vol): problems.append(f"level {l} volumes")
                if n_levels > 1 and not np.array_equal(L["coords"], coords): problems.append(f"level {l} coords")
            ours.close(); oc.close()
            print(f"seed {seed}: {kind} {name} {[l.nel for l in mg.levels]}: " + ("ok" if not problems else "MISMATCH " + "; ".join(problems)), flush=True)
            bad += 1 if problems else 0
        finally:
            shutil.rmtree(d, ignore_errors=True)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
