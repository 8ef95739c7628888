"""fuzz_partitioned seed 2070 (tet rotor37 260 nodes, 3 parts rcb, variant 32): which nodes differ between the half-row kernel and the
node gather on the SAME partitioned solver, first fused stage (ghosts current, no exchange needed)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("mg-cfd-app-plain_amd", "", "tests", "oracle", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))
import fuzz_parity, mgcfd
from conftest import perturbed_state
from mgcfd.partition import partition_level, rcb_partition, slab_partition
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2070
rng = np.random.default_rng(5000 + seed)
while True:
    kind, name, mg, _ = fuzz_parity.make_case(rng)
    if name != "fvcorr" and 20 <= mg.levels[0].nel <= 5000:
        break
mg.levels = mg.levels[:1]; mg.levels[0].mg_map = None
n_parts = int(rng.integers(2, 6)); partitioner = str(rng.choice(["slab", "rcb"])); fused = bool(rng.integers(2)); variant = int(rng.choice([-1, 0, 1, 2, 3, 16, 32]))
print(kind, name, mg.levels[0].nel, n_parts, partitioner, fused, variant)
L = mgcfd.generated_to_levels(mg)[0]
parts = partition_level(L, (slab_partition if partitioner == "slab" else rcb_partition)(np.asarray(L["coords"]), n_parts))
whole = mgcfd.Solver.from_arrays([L], mg.mesh_variant)
q0 = perturbed_state(L["nel"], whole.far_field()[:5], seed=seed)
whole.close()
for P in parts:
    out = {}
    for v in (0, variant):
        s = mgcfd.Solver.from_arrays([P.level], mg.mesh_variant, n_owned=[P.n_owned])
        s.set_option("flux_variant", v)
        s.set(0, "variables", q0[P.global_ids])
        print(f"part {P.rank}: owned {P.n_owned} local {P.n_local} half rows available: {s.has_half_rows(0)} tiling {s.tiling(0)}") if v == 0 else None
        # the flux alone first
        s.zero_fluxes(0); s.compute_fluxes(0)
        out[v, "fluxes"] = s.get(0, "fluxes")
        s.zero_fluxes(0)
        s.sweep_begin(0); s.sweep_stage(0, 0, False)
        out[v, "stage"] = s.get(0, "stage")
        s.close()
    for what in ("fluxes", "stage"):
        a, b = out[0, what][:P.n_owned], out[variant, what][:P.n_owned]
        bad = np.argwhere(a.view(np.int64) != b.view(np.int64))
        print(f"  part {P.rank} {what}: {len(set(bad[:, 0]))} owned nodes differ", (sorted(set(bad[:, 0]))[:12] if len(bad) else ""))
        if len(bad):
            n = bad[0, 0]
            deg = int(((P.level['edges']['a'][:P.level['n_internal']] == n) | (P.level['edges']['b'][:P.level['n_internal']] == n)).sum())
            print(f"    node {n}: degree {deg}; node gather {a[n]}, variant {variant} {b[n]}")
