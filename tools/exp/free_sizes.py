#!/usr/bin/env python3
"""compute_flux_edge launch by level size: bit-identical, contracted and order-free kernels (event-timed batches)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
import bench, mgcfd
for n in [int(x) for x in (sys.argv[1:] or ["58", "67", "84", "96", "134"])]:
    mg, levels = bench.build_workload(n)
    s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
    nel, ni = s.nel(0), s.num_internal_edges(0)
    s.set(0, "variables", bench.perturbed_state(nel, s.far_field()[:5]))
    algo = 40 * ni + 80 * nel
    out = []
    for name, exact, v in (("exact", 1, 1), ("contracted", 0, 1), ("order-free", 0, 65)):
        s.set_option("exact", exact); s.set_option("flux_variant", v)
        s.bench_flux(0, 300)
        t = statistics.median(s.bench_flux(0, 300) for _ in range(3))
        out.append(f"{name} {t*1e6:7.2f} us frac {algo/t/8e12:.3f}")
    print(f"{n}^3: {nel} nodes {ni} edges {-(-nel//256)} tiles {algo/1e6:.1f} MB | " + " | ".join(out), flush=True)
    s.close()
