#!/usr/bin/env python3
"""Sweep and flux-kernel rates on an UNSTRUCTURED level (Delaunay tetrahedra, median-dual metrics,
meshgen.make_tet_level): the same measurements bench.py takes on the lattice workload, to show how the tile
kernels behave with 7.7 edges per node, degrees up to ~55 and halos larger than a tile holds.
    python tools/tet_mesh_bench.py [--nodes 120000] [--steps 300]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd"))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=120000)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--variants", default="-1,1,2")
    ap.add_argument("--fast", action="store_true", help="MGCFD_OPT_EXACT = 0 for every variant (auto then picks the order-free kernel where it wins)")
    ap.add_argument("--vcycle", default="", help="comma-separated node counts of a hierarchy: also time a multigrid cycle on it")
    args = ap.parse_args()
    import numpy as np, torch, mgcfd
    from mgcfd import meshgen
    from bench import perturbed_state
    t0 = time.time()
    mg = meshgen.MultigridMesh(mesh_name="m6wing")
    mg.levels.append(meshgen.make_tet_level(args.nodes, seed=0))
    levels = mgcfd.generated_to_levels(mg)
    print(f"mesh built in {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
    out = []
    for variant in [int(v) for v in args.variants.split(",")]:
        s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
        stream = torch.cuda.Stream()
        torch.cuda.set_stream(stream)
        s.set_stream(stream.cuda_stream)
        if variant >= 64 or args.fast:      # bit 6: the order-free kernel, which lives in the contracted namespace
            s.set_option("exact", 0)
        s.set_option("flux_variant", variant)
        nel, E = s.nel(0), s.num_internal_edges(0)
        s.set(0, "variables", perturbed_state(nel, s.far_field()[:5]))
        for _ in range(50):
            s.smooth(0, 1)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(args.steps):
            s.smooth(0, 1)
        torch.cuda.synchronize()
        sweep = (time.perf_counter() - t) / args.steps
        flux = s.bench_flux(0, 50)
        rc, _ = s.check_for_invalid_variables(0)
        out.append({"variant": variant, "nodes": nel, "edges": E, "edge_once": bool(s.has_edge_once(0)), "order_free": bool(s.has_order_free(0)) and variant >= 64,
                    "sweep_us": round(sweep * 1e6, 2), "sweep_gedges_s": round(3 * E / sweep / 1e9, 2),
                    "flux_us": round(flux * 1e6, 2), "flux_gedges_s": round(E / flux / 1e9, 2),
                    "flux_roofline_frac": round((40 * E + 80 * nel) / flux / 8e12, 4), "state_valid": rc == 0})
        s.close()
    if args.vcycle:
        sizes = [int(v) for v in args.vcycle.split(",")]
        mgh = meshgen.make_tet_multigrid(sizes, "m6wing", seed=0)
        s = mgcfd.Solver.from_arrays(mgcfd.generated_to_levels(mgh), mgh.mesh_variant)
        if args.fast:
            s.set_option("exact", 0)
        stream = torch.cuda.Stream()
        torch.cuda.set_stream(stream)
        s.set_stream(stream.cuda_stream)
        s.run_cycles(5)
        torch.cuda.synchronize()
        t = time.perf_counter()
        s.run_cycles(50)
        torch.cuda.synchronize()
        out.append({"vcycle_levels": sizes, "ms_per_cycle": round((time.perf_counter() - t) / 50 * 1e3, 4),
                    "tiling": [s.tiling(l) for l in range(len(sizes))]})
        s.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
