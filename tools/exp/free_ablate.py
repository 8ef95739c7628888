#!/usr/bin/env python3
"""Event-timed batches of the order-free flux kernel from experiment builds (tools/exp_build.py): MGCFD_LIB=... python tools/exp/free_ablate.py"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
import bench, mgcfd
mg, levels = bench.build_workload(int(sys.argv[1]) if len(sys.argv) > 1 else 67)
s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
s.set(0, "variables", bench.perturbed_state(s.nel(0), s.far_field()[:5]))
variant = int(os.environ.get("FREE_VARIANT", "65"))
s.set_option("exact", 0); s.set_option("flux_variant", variant)
s.bench_flux(0, 2000)
ts = [s.bench_flux(0, 500) for _ in range(5)]
import numpy as np
q0 = bench.perturbed_state(s.nel(0), s.far_field()[:5])
s.set(0, "variables", q0); s.zero_fluxes(0); s.compute_fluxes(0); f = s.get(0, "fluxes").copy()
s.set_option("exact", 1); s.set_option("flux_variant", 1); s.zero_fluxes(0); s.compute_fluxes(0); fr = s.get(0, "fluxes")
print(os.environ.get("MGCFD_LIB", "default build"), f"variant {variant}: median {statistics.median(ts)*1e6:.2f} us; one launch vs the bit-identical kernel: {np.abs(f - fr).max() / np.abs(fr).max():.3e}")
