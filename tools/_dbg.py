import sys, os
sys.path.insert(0, "mg-cfd-app-plain_amd"); sys.path.insert(0, "tests")
import numpy as np, mgcfd
from mgcfd import meshgen
from mgcfd.partition import partition_level, slab_partition
from conftest import perturbed_state
mg = meshgen.make_multigrid((14,), "m6wing", seed=4, cavity_radius=0.15, jitter=0.2, area_noise=0.05, volume_noise=0.05)
L = mgcfd.generated_to_levels(mg)[0]
parts = partition_level(L, slab_partition(np.asarray(L["coords"]), 3))
P = parts[0]
for v in (0, 2):
    s = mgcfd.Solver.from_arrays([P.level], mg.mesh_variant, n_owned=[P.n_owned])
    s.set_option("flux_variant", v)
    q = perturbed_state(P.level["nel"], s.far_field()[:5], seed=9)
    s.set(0, "variables", q)
    def chk(tag):
        for a in ("variables", "fluxes", "step_factors", "old_variables"):
            x = s.get(0, a)
            print(v, tag, a, "nan:", int(np.isnan(x).sum()), "absmax", float(np.nanmax(np.abs(x))))
    s.copy_old_variables(0); s.step_factor_local(0); s.step_factor_apply(0)
    chk("after sf")
    for j in range(3):
        s.compute_fluxes(0); chk(f"flux {j}")
        s.time_step(0, j); chk(f"ts {j}")
    s.close()
