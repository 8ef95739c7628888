#!/usr/bin/env python3
"""Randomised parity sweep (GPU box): random hierarchies of every generator kind — lattices, Delaunay tetrahedra,
hubs, random graphs — with random sizes, mesh names and options, each a few cycles against the oracle, bit for bit.
    python tools/fuzz_parity.py [--seeds 40] [--first 0]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))


def make_case(rng):
    import numpy as np
    from mgcfd import meshgen
    kind = rng.choice(["lattice", "tet", "hub", "graph", "tet", "tet"])
    name = str(rng.choice(["m6wing", "rotor37", "la_cascade", "fvcorr"]))
    if kind == "lattice":
        n0 = int(rng.integers(3, 22))
        sizes = [n0] + [max(2, n0 // (2 ** k)) for k in range(1, int(rng.integers(1, 4)))]
        if name == "fvcorr":
            sizes = sizes[:1]
        mg = meshgen.make_multigrid(tuple(sizes), name, seed=int(rng.integers(1 << 30)), jitter=0.2, area_noise=0.05, volume_noise=0.05,
                                    cavity_radius=float(rng.choice([0.0, 0.2])))
        cycles = 3
    elif kind == "tet":
        n0 = int(rng.integers(40, 9000))
        sizes = [n0] + [max(12, n0 // (4 ** k)) for k in range(1, int(rng.integers(1, 4)))]
        if name == "fvcorr":
            sizes = sizes[:1]
        mg = meshgen.make_tet_multigrid(tuple(sizes), name, seed=int(rng.integers(1 << 30)))
        cycles = 3
    elif kind == "hub":
        mg = meshgen.MultigridMesh(mesh_name=name)
        mg.levels.append(meshgen.make_hub_level(int(rng.integers(1, 3000)), scale=1e-6, seed=int(rng.integers(1 << 30))))
        cycles = 2
    else:
        mg = meshgen.MultigridMesh(mesh_name=name)
        mg.levels.append(meshgen.make_random_graph_level(int(rng.integers(50, 6000)), degree=int(rng.integers(2, 24)), seed=int(rng.integers(1 << 30))))
        if rng.random() < 0.5 and name != "fvcorr":
            mg.levels.append(meshgen.make_random_graph_level(int(rng.integers(10, 500)), degree=int(rng.integers(2, 10)), seed=int(rng.integers(1 << 30))))
            mg.levels[0].mg_map = rng.integers(0, mg.levels[1].nel, mg.levels[0].nel).astype(np.int64)
        cycles = 2
    return kind, name, mg, cycles


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=40)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--fast", action="store_true", help="MGCFD_OPT_EXACT=0 (FMA contraction): compare by the reference's -v tolerance rule instead of bit for bit")
    args = ap.parse_args()
    import numpy as np
    import mgcfd
    import oracle_py as oracle
    import test_gpu_parity as T
    bad = 0
    for seed in range(args.first, args.first + args.seeds):
        rng = np.random.default_rng(seed)
        kind, name, mg, cycles = make_case(rng)
        levels = mgcfd.generated_to_levels(mg)
        try:
            want, want_rms = T._oracle_solve_arrays(oracle, levels, mg.mesh_variant, cycles)
        except AssertionError:
            print(f"seed {seed}: {kind} {name} {[l.nel for l in mg.levels]}: the oracle's state goes invalid, skipped", flush=True)
            continue
        opts = {"fuse_update": int(rng.integers(0, 2)), "flux_variant": int(rng.choice([-1, 0, 1, 2, 3, 16, 32, 33])), "graph": int(rng.integers(0, 2))}
        s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
        for k, v in opts.items():
            s.set_option(k, v)
        if args.fast:
            s.set_option("exact", 0)
        rms = s.run_cycles(cycles)
        if args.fast:
            # the reference's own -v tolerance rule (validation.cpp:140-199): -1 = no value out of tolerance
            lib = oracle.load()
            ok = all(lib.ora_identify_differences(oracle.ptr(np.ascontiguousarray(s.get(l, "variables"))), oracle.ptr(np.ascontiguousarray(want[l])),
                                                  levels[l]["nel"], mg.mesh_variant) == -1 for l in range(len(levels)))
        else:
            ok = all(np.array_equal(s.get(l, "variables").view(np.int64), want[l].view(np.int64)) for l in range(len(levels)))
        # (the RMS history: 1e-12 where the state is bit-identical — a tree sum against the reference's loop —, north_star's 1e-10 in the
        #  fast mode, whose state is a few 1e-12 away after three cycles: seed 41093 reads 2.6e-12 with every tile order)
        ok = ok and np.allclose(rms, want_rms, rtol=1e-10 if args.fast else 1e-12, atol=1e-300)
        t = s.tiling(0)
        s.close()
        print(f"seed {seed}: {kind} {name} {[l.nel for l in mg.levels]} {opts} list={t['list_entries']} overflow={t['overflow_refs']}: {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += 0 if ok else 1
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
