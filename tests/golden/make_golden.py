#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ with the REAL reference.

Needs oracle/_ref/ (built by oracle/build_ref.sh from /root/reference/src — only possible in
the build container).  Everything written here is DATA: synthetic input files in the
reference's formats (produced by mgcfd.meshgen, ours) and the outputs the reference computed
for them:
  <case>/input/…                    input.dat, mesh, .coords, MG-map files
  <case>/variables.level0.txt       the reference binary's `--output-variables` dump (%.17e)
  <case>/LoopNumIters.csv           the reference binary's iteration counts
  <case>/stdout.txt                 its progress lines (RMS per cycle)
  <case>/kernels.npz                per-kernel vectors from the reference kernels called through
                                    oracle/ref_harness.cpp on a seeded perturbed state
The reference repository ships no golden vectors for this path, so these are the pins.
Run:  python tests/golden/make_golden.py
"""
import ctypes as C
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_py as O  # noqa: E402
from mgcfd import meshgen  # noqa: E402

CASES = {
    # name: (sizes, mesh_name, generator kwargs, cycles, duplicate)
    "m6_2lvl": ((7, 4), "m6wing", dict(seed=3, cavity_radius=0.2, jitter=0.2, area_noise=0.05, volume_noise=0.05), 3, 1),
    "m6_3lvl": ((9, 6, 4), "m6wing", dict(seed=11, cavity_radius=0.15, jitter=0.25, area_noise=0.08, volume_noise=0.1), 2, 1),
    "m6_2lvl_dup2": ((7, 4), "m6wing", dict(seed=5, cavity_radius=0.2, jitter=0.1, area_noise=0.03, volume_noise=0.02), 2, 2),
    "fvcorr_1lvl": ((8,), "fvcorr", dict(seed=7, cavity_radius=0.01, volume_noise=0.02), 20, 1),
    # run with the reference built with -DLEGACY_ORDERING (only the binary's outputs; undamped, so the order matters)
    "fvcorr_1lvl_legacy_ordering": ((8,), "fvcorr", dict(seed=8, cavity_radius=0.01, volume_noise=0.02), 30, 1),
    # unstructured: Delaunay tetrahedra with median-dual metrics (sizes = node counts), degrees 4..30, wall + far-field hull faces
    "tet_2lvl": ((420, 90), "rotor37", dict(seed=4, tet=True), 4, 1),
    # mixed element types on level 0 (hexahedral core, prism layers on a wall, tetrahedral far field: internal degrees 3..14)
    "mixed_2lvl": ((9, 5), "m6wing", dict(seed=6, jitter=0.2, area_noise=0.05, volume_noise=0.05, mixed=True), 3, 1),
    # runs that the reference ABORTS (check_for_invalid_variables, validation.cpp:107-138): a 300-spoke hub with
    # non-physical weights, undamped; only the binary's stdout (cycle lines, ERROR line, first "Cell" line) and exit code
    "fvcorr_hub_nan": ((300,), "fvcorr", dict(seed=5, hub=1.2e-4), 40, 1),                   # NaN in cycle 7
    "fvcorr_hub_negative_energy": ((300,), "fvcorr", dict(seed=5, hub=1e-4), 40, 1),         # density*energy < 0 in cycle 32
}


def perturbed_state(nel, ff_var, seed, amplitude=0.01):
    rng = np.random.default_rng(seed)
    base = np.tile(np.asarray(ff_var, dtype=np.float64), (nel, 1))
    base[:, 2:4] += 0.3
    return base * (1.0 + amplitude * rng.uniform(-1.0, 1.0, base.shape))


def kernel_vectors(ref, case_dir, info, mesh_variant):
    """Call the reference kernels on seeded inputs, level by level."""
    n_levels = info["num_levels"]
    ref.ref_init(n_levels, mesh_variant)
    ff = np.zeros(17)
    ref.ref_get_farfield(O.ptr(ff))
    out = {"far_field": ff}
    levels = []
    for l in range(n_levels):
        path = os.path.join(case_dir, "input", info["levels"][l]).encode()
        sizes = np.zeros(8, dtype=np.int64)
        ref.ref_read_grid(path, O.ptr(sizes))
        nel, n_edges = int(sizes[0]), int(sizes[1])
        vol = np.zeros(nel)
        edges = np.zeros(n_edges, dtype=O.EDGE_DTYPE)
        coords = np.zeros((nel, 3))
        ref.ref_grid_copy(O.ptr(vol), O.ptr(edges), O.ptr(coords))
        out[f"L{l}_sizes"] = sizes
        out[f"L{l}_edges_raw"] = edges.copy()
        out[f"L{l}_volumes"] = vol
        if mesh_variant != 0:
            ref.ref_adjust_ewt(O.ptr(coords), n_edges, O.ptr(edges))
            # damping per mesh (src/euler3d_cpu_double.cpp:337-352): m6wing 5e-8, la_cascade 1e-7, rotor37 2e-7
            ref.ref_dampen_ewt(n_edges, O.ptr(edges), {2: 5e-8, 3: 1e-7, 4: 2e-7}[mesh_variant])
        out[f"L{l}_edges"] = edges.copy()
        levels.append((nel, sizes, vol, edges, coords))
    for l, (nel, sizes, vol, edges, coords) in enumerate(levels):
        ref.ref_set_level(l)
        n_int, n_bnd, n_wall = int(sizes[2]), int(sizes[3]), int(sizes[4])
        q = perturbed_state(nel, ff[:5], seed=100 + l)
        out[f"L{l}_q"] = q
        f = np.random.default_rng(7 + l).normal(size=(nel, 5))
        out[f"L{l}_flux_in"] = f.copy()
        ref.ref_compute_flux_edge(int(sizes[5]), n_int, O.ptr(edges), O.ptr(q), O.ptr(f))
        out[f"L{l}_flux_internal"] = f.copy()
        ref.ref_compute_boundary_flux_edge(int(sizes[6]), n_bnd, O.ptr(edges), O.ptr(q), O.ptr(f))
        out[f"L{l}_flux_boundary"] = f.copy()
        ref.ref_compute_wall_flux_edge(int(sizes[7]), n_wall, O.ptr(edges), O.ptr(q), O.ptr(f))
        out[f"L{l}_flux_wall"] = f.copy()
        g = np.zeros((nel, 5))
        ref.ref_indirect_rw(int(sizes[5]), n_int, O.ptr(edges), O.ptr(q), O.ptr(g))
        out[f"L{l}_indirect_rw"] = g
        sf = np.zeros(nel)
        if mesh_variant == 0:
            ref.ref_compute_step_factor_legacy(nel, O.ptr(q), O.ptr(vol), O.ptr(sf))
        else:
            ref.ref_compute_step_factor(nel, O.ptr(q), O.ptr(vol), O.ptr(sf))
        out[f"L{l}_step_factors"] = sf
        rng = np.random.default_rng(5 + l)
        flux = rng.normal(size=(nel, 5)) * 1e-3
        old = q * (1.0 + 1e-3 * rng.uniform(-1, 1, q.shape))
        out[f"L{l}_ts_flux"] = flux
        out[f"L{l}_ts_old"] = old
        for j in range(3):
            fj, vj = flux.copy(), np.zeros_like(q)
            ref.ref_time_step(j, nel, O.ptr(sf), O.ptr(fj), O.ptr(old), O.ptr(vj))
            out[f"L{l}_ts_j{j}"] = vj
        res = np.zeros_like(q)
        ref.ref_residual(nel, O.ptr(old), O.ptr(q), O.ptr(res))
        out[f"L{l}_residual"] = res
        out[f"L{l}_rms"] = np.array([ref.ref_calc_rms(nel, O.ptr(res))])
    for l in range(n_levels - 1):
        nel_f, sizes_f, _, edges_f, coords_f = levels[l]
        nel_c, _, _, _, coords_c = levels[l + 1]
        mapping = np.loadtxt(os.path.join(case_dir, "input", info["mg_mapping"][l]), dtype=np.int64)[1:].copy()
        qf = perturbed_state(nel_f, ff[:5], seed=300 + l)
        qc = perturbed_state(nel_c, ff[:5], seed=310 + l)
        out[f"T{l}_qf"], out[f"T{l}_qc"] = qf.copy(), qc.copy()
        scratch = np.zeros(max(nel_c, 1), dtype=np.int64)
        ref.ref_set_level(l + 1)
        ref.ref_mg_restrict(O.ptr(qf), O.ptr(qc), nel_c, O.ptr(mapping), O.ptr(scratch), len(mapping))
        out[f"T{l}_restrict"] = qc.copy()
        rng = np.random.default_rng(320 + l)
        r_c = rng.normal(size=(nel_c, 5)) * 1e-4
        r_f = rng.normal(size=(nel_f, 5)) * 1e-4
        out[f"T{l}_rc"], out[f"T{l}_rf"] = r_c, r_f
        v = out[f"T{l}_qf"].copy()
        ref.ref_set_level(l)
        ref.ref_prolong_residuals_interpolate_proper(O.ptr(edges_f), int(sizes_f[2]), O.ptr(r_c), O.ptr(r_f), O.ptr(v), nel_f,
                                                     O.ptr(mapping), O.ptr(np.ascontiguousarray(coords_c)), O.ptr(np.ascontiguousarray(coords_f)))
        out[f"T{l}_prolong"] = v
    return out


def main():
    if not O.have_reference():
        raise SystemExit("oracle/_ref is missing: run oracle/build_ref.sh in the build container first")
    ref = O.load_reference()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for name, (sizes, mesh_name, kw, cycles, dup) in CASES.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:      # python make_golden.py [case ...]
            continue
        d = os.path.join(HERE, name)
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(os.path.join(d, "input"))
        kw = dict(kw)
        hub = kw.pop("hub", None)
        if hub is not None:
            mg = meshgen.MultigridMesh(mesh_name=mesh_name)
            mg.levels.append(meshgen.make_hub_level(sizes[0], scale=hub, **kw))
        else:
            make = meshgen.make_tet_multigrid if kw.pop("tet", False) else (meshgen.make_mixed_multigrid if kw.pop("mixed", False) else meshgen.make_multigrid)
            mg = make(sizes, mesh_name, **kw)
        meshgen.write_input(mg, os.path.join(d, "input"))
        out_dir = os.path.join(d, "_out")
        os.makedirs(out_dir)
        legacy = name.endswith("legacy_ordering")
        cmd = [O.REF_BIN_LEGACY if legacy else O.REF_BIN, "-i", "input.dat", "-d", os.path.join(d, "input"), "-o", out_dir + "/", "-g", str(cycles),
               "-m", str(dup), "--output-variables"]
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, check=hub is None)
        if hub is not None:
            assert r.returncode != 0, "this case is meant to be aborted by check_for_invalid_variables"
            with open(os.path.join(d, "stdout.txt"), "w") as f:
                f.write("\n".join(l for l in r.stdout.splitlines() if "cycle" in l.lower() or "ERROR" in l or l.startswith("Cell")) + "\n")
            with open(os.path.join(d, "case.txt"), "w") as f:
                f.write(f"cycles = {cycles}\nduplicate = {dup}\nmesh_name = {mesh_name}\nreturncode = {r.returncode}\n")
            shutil.rmtree(out_dir)
            print(f"{name}: aborted by the reference with exit code {r.returncode}")
            continue
        shutil.copy(os.path.join(out_dir, f"variables.size={dup}x.cycles={cycles}.level=0"), os.path.join(d, "variables.level0.txt"))
        shutil.copy(os.path.join(out_dir, "LoopNumIters.csv"), os.path.join(d, "LoopNumIters.csv"))
        with open(os.path.join(d, "stdout.txt"), "w") as f:
            f.write("\n".join(l for l in r.stdout.splitlines() if "cycle" in l.lower()) + "\n")
        with open(os.path.join(d, "case.txt"), "w") as f:
            f.write(f"cycles = {cycles}\nduplicate = {dup}\nmesh_name = {mesh_name}\n")
            if name.endswith("legacy_ordering"):
                f.write("legacy_ordering = 1\n")
        shutil.rmtree(out_dir)
        if not legacy:
            info = O.parse_input_dat(os.path.join(d, "input", "input.dat"))
            vec = kernel_vectors(ref, d, info, info["mesh_variant"])
            np.savez_compressed(os.path.join(d, "kernels.npz"), **vec)
        print(f"{name}: {[l.nel for l in mg.levels]} nodes, {cycles} cycles, x{dup}")


if __name__ == "__main__":
    main()
