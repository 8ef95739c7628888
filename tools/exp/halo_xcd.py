import sys, os
sys.path.insert(0,'mg-cfd-app-plain_amd'); sys.path.insert(0,'.')
import bench, mgcfd
mesh = sys.argv[1] if len(sys.argv) > 1 else "lattice"
mg, levels = bench.build_workload(67, mesh=mesh)
s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant); s.close()
