#!/usr/bin/env python3
"""The kernels of a PARTITIONED sweep, for a rocprofv3 kernel-trace summary (profiles/r2_group8_kernel_stats.csv): the 134^3
level (2.4 M nodes / 7.2 M edges) split into 8 parts = 8 solvers of this process on ONE GPU (mgcfd_group_*: boundary tiles,
one launch that stores the message into the neighbours' ghost slots, interior tiles; the time step's minimum read from the
peers), `--sweeps` sweeps issued by one host thread.  Eight ranks share the device here, so the SUM of a rank's kernel
durations is what a rank's GPU would spend per sweep — not the wall time of this run.
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/exp/group8_profile.py [--parts 8] [--sweeps 40]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("MGCFD_GROUP_THREADS", "0")           # one thread issues every rank's launches
import bench, mgcfd
from mgcfd.partition import partition_level, rcb_partition

ap = argparse.ArgumentParser()
ap.add_argument("--parts", type=int, default=8)
ap.add_argument("--sweeps", type=int, default=40)
ap.add_argument("--lattice", type=int, default=bench.LATTICE_8X)
a = ap.parse_args()
mg, levels = bench.build_workload(a.lattice)
L = levels[0]
parts = partition_level(L, rcb_partition(np.asarray(L["coords"]), a.parts))
solvers = []
q0 = None
for P in parts:
    s = mgcfd.Solver.from_arrays([P.level], mg.mesh_variant, n_owned=[P.n_owned])
    if q0 is None:
        q0 = bench.perturbed_state(L["nel"], s.far_field()[:5])
    s.set(0, "variables", q0[P.global_ids])
    solvers.append(s)
g = mgcfd.Group(solvers)
for P, s in zip(parts, solvers):
    s.rank_set_halo(0, P)
g.exchange(0)
g.sweeps(0, a.sweeps)
g.synchronize()
info = solvers[0].rank_halo_info(0)
print(f"{a.parts} parts of {L['nel']} nodes, {a.sweeps} sweeps; rank 0: {parts[0].n_owned} owned nodes, {info}; rms {g.rms(0):.6e}")
