#!/usr/bin/env python3
"""Launch only the flux-gather kernel N times on the bench workload (for rocprofv3 --pmc runs)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
import bench, mgcfd
ap = argparse.ArgumentParser()
ap.add_argument("--lattice", type=int, default=67)
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--launches", type=int, default=20)
ap.add_argument("--fast", action="store_true")
ap.add_argument("--tet", type=int, default=0, help="node count of a Delaunay tetrahedral level to use instead of the lattice")
ap.add_argument("--mesh", default="lattice", choices=["lattice", "mixed"], help="bench.py --mesh")
a = ap.parse_args()
if a.tet:
    from mgcfd import meshgen
    mg = meshgen.MultigridMesh(mesh_name="m6wing")
    mg.levels.append(meshgen.make_tet_level(a.tet, seed=0))
    levels = mgcfd.generated_to_levels(mg)
else:
    mg, levels = bench.build_workload(a.lattice, mesh=a.mesh)
s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
s.set(0, "variables", bench.perturbed_state(s.nel(0), s.far_field()[:5]))
s.set_option("exact", 0 if a.fast else 1)
s.set_option("flux_variant", a.variant)
print("avg us", s.bench_flux(0, a.launches) * 1e6)
