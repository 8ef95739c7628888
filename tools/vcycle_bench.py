#!/usr/bin/env python3
"""BASELINE.json configs[2] in its synthetic form: 4-level M6-like hierarchy
(67^3/55^3/48^3/43^3 lattices = 300,763/166,375/110,592/79,507 nodes), mesh_name = m6wing, V-cycles on
one MI355X.  Prints seconds per MG cycle and the per-loop edge/node rates."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd"))
import mgcfd
from mgcfd import meshgen
ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="67,55,48,43")
ap.add_argument("--cycles", type=int, default=25)
ap.add_argument("--repeats", type=int, default=3)
ap.add_argument("--fast", action="store_true")
ap.add_argument("--variant", type=int, default=-1, help="MGCFD_OPT_FLUX_VARIANT")
ap.add_argument("--timers", action="store_true", help="per-loop hipEvent timing (unfused, as the driver's default)")
a = ap.parse_args()
sizes = tuple(int(x) for x in a.sizes.split(","))
t0 = time.time()
mg = meshgen.make_multigrid(sizes, "m6wing", seed=0, jitter=0.2, area_noise=0.02, volume_noise=0.02)
levels = mgcfd.generated_to_levels(mg)
t1 = time.time()
s = mgcfd.Solver.from_arrays(levels, mg.mesh_variant)
t2 = time.time()
s.set_option("exact", 0 if a.fast else 1)
s.set_option("timing", 1 if a.timers else 0)
s.set_option("flux_variant", a.variant)
s.run_cycles(2)
best = 1e9
for _ in range(a.repeats):
    s.reset_monitoring()
    t = time.perf_counter()
    rms = s.run_cycles(a.cycles)
    best = min(best, time.perf_counter() - t)
edge_iters = sum(s.loop_iters(l)["flux"] for l in range(s.num_levels))
out = {"workload": f"M6-like 4-level synthetic hierarchy {[l.nel for l in mg.levels]} nodes", "cycles": a.cycles,
       "seconds_per_cycle": best / a.cycles, "flux_edge_iterations_per_cycle": edge_iters // a.cycles,
       "whole_cycle_medges_per_s": edge_iters / best / 1e6, "rms_last": float(rms[-1]),
       "mesh_build_s": round(t1 - t0, 1), "plan_and_upload_s": round(t2 - t1, 1)}
if a.timers:
    out["loop_times_s"] = {l: {k: round(v, 6) for k, v in s.loop_times(l).items() if v} for l in range(s.num_levels)}
print(json.dumps(out))
