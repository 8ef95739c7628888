"""Host-only bounds audit of the gather plans (mgcfd_plan_audit, preprocess.cpp: audit_level_plan): for every kind of level
the generators make — lattices with cavities, the 13^3 level of the sharded-sweep test (9 tiles, a 130-node last tile:
the level of round 1's abort), random graphs, Delaunay tetrahedra, a hub, partitioned levels with ghosts,
whole hierarchies with their transfer plans — every index a kernel forms from the plan's tables must lie inside the array
it indexes.  No GPU."""
import numpy as np
import pytest

import mgcfd
from mgcfd import meshgen


def _levels(mg):
    return mgcfd.generated_to_levels(mg)


def _single(level, name="m6wing"):
    mg = meshgen.MultigridMesh(mesh_name=name)
    mg.levels.append(level)
    return mg


CASES = {
    "lattice 13^3 with a cavity (the sharded-sweep level)": lambda: meshgen.make_multigrid((13,), "m6wing", seed=7, cavity_radius=0.12, jitter=0.2, area_noise=0.05, volume_noise=0.05),
    "three lattice levels": lambda: meshgen.make_multigrid((13, 9, 6), "m6wing", seed=11, cavity_radius=0.12, jitter=0.25, area_noise=0.08, volume_noise=0.1),
    "fvcorr box": lambda: meshgen.make_multigrid((12,), "fvcorr", seed=5, cavity_radius=0.01, volume_noise=0.02),
    "random graph, degree 10 (long rows)": lambda: _single(meshgen.make_random_graph_level(3000, degree=10, seed=4)),
    "tetrahedra, two levels": lambda: meshgen.make_tet_multigrid((4000, 700), "m6wing", seed=3),
    "hub of 700 spokes": lambda: _single(meshgen.make_hub_level(700, scale=1e-4, seed=6)),
    "lattice 24^3 (54 tiles)": lambda: meshgen.make_multigrid((24,), "m6wing", seed=0, jitter=0.2, area_noise=0.02, volume_noise=0.02, permute=True),
}


@pytest.mark.parametrize("name", list(CASES))
def test_every_plan_index_is_in_range(name):
    mg = CASES[name]()
    report = mgcfd.plan_audit(_levels(mg), mg.mesh_variant)
    assert report == "", f"{name}:\n{report}"


@pytest.mark.parametrize("tile_order,tile_curve", [("2", "1"), ("1", "2"), ("2", "0")])
def test_levels_of_more_tiles_than_one_round_are_in_range_with_their_tiles_reordered(tile_order, tile_curve, monkeypatch):
    """A level of more than 768 tiles has the cheapest tiles of every XCD's range moved to the range's end (preprocess.cpp,
    "dispatch order of the tiles"; MGCFD_TILE_ORDER=1: the whole range by cost): the permutation must stay one — every node
    once — and every index of the plan in range, on the mixed-element level (rows of 3 ... 14) and a partitioned lattice."""
    from mgcfd.partition import partition_level, rcb_partition
    monkeypatch.setenv("MGCFD_TILE_ORDER", tile_order)
    monkeypatch.setenv("MGCFD_TILE_CURVE", tile_curve)      # (the tiles along a Morton / Hilbert curve first: the default / 2)
    mg = meshgen.make_mixed_multigrid((60,), "m6wing", seed=2, jitter=0.2, area_noise=0.02, volume_noise=0.02, permute=True)
    levels = _levels(mg)
    assert levels[0]["nel"] > 768 * 256
    report = mgcfd.plan_audit(levels, mg.mesh_variant)
    assert report == "", report
    P = partition_level(levels[0], rcb_partition(np.asarray(levels[0]["coords"]), 1))[0]
    cut = levels[0]["nel"] - 5000                          # (the last 5,000 nodes as ghosts: complete owned tiles only may move)
    report = mgcfd.plan_audit([P.level], mg.mesh_variant, n_owned=[cut])
    assert report == "", report


def test_partitioned_levels_and_hierarchies_are_in_range():
    from mgcfd.partition import partition_hierarchy, partition_level, rcb_partition
    mg = meshgen.make_multigrid((14, 9), "m6wing", seed=2, cavity_radius=0.1, jitter=0.2, area_noise=0.05, volume_noise=0.05)
    levels = _levels(mg)
    part = rcb_partition(np.asarray(levels[0]["coords"]), 3)
    for P in partition_level(levels[0], part):
        report = mgcfd.plan_audit([P.level], mg.mesh_variant, n_owned=[P.n_owned])
        assert report == "", report
    for H in partition_hierarchy(levels, part):
        lv, owned, keys = H.solver_args()
        report = mgcfd.plan_audit(lv, mg.mesh_variant, n_owned=owned, order_keys=keys)
        assert report == "", report


def test_the_audit_sees_a_broken_plan():
    """A multigrid map that points outside the coarse level is refused when the plan is built; an internal edge that names a
    node outside the level must not pass silently either."""
    mg = meshgen.make_multigrid((9, 5), "m6wing", seed=3, cavity_radius=0.15, jitter=0.2)
    levels = _levels(mg)
    bad = [dict(L) for L in levels]
    m = np.array(bad[0]["mg_map"], dtype=np.int64).copy()
    m[5] = bad[1]["nel"] + 3
    bad[0]["mg_map"] = m
    with pytest.raises(mgcfd.MgcfdError):
        mgcfd.plan_audit(bad, mg.mesh_variant)


@pytest.mark.parametrize("name", ["lattice 24^3 (54 tiles)", "tetrahedra, two levels", "three lattice levels", "random graph, degree 10 (long rows)", "lattice 40^3 (250 tiles)", "mixed elements 36^3"])
def test_the_plan_does_not_depend_on_the_number_of_host_threads(name, monkeypatch):
    """Round 4 builds the per-tile part of a plan (halo lists, tile-local codes, edge-once lists, half rows) on several host
    threads, each over a contiguous range of tiles, and joins the parts in tile order: every array and every counter must come
    out the same as from one thread (a digest over all of them, mgcfd_plan_audit with MGCFD_PLAN_DIGEST)."""
    more = {"lattice 40^3 (250 tiles)": lambda: meshgen.make_multigrid((40,), "m6wing", seed=1, jitter=0.2, area_noise=0.02, volume_noise=0.02, permute=True),
            "mixed elements 36^3": lambda: meshgen.make_mixed_multigrid((36,), "m6wing", seed=2, jitter=0.2, area_noise=0.02, volume_noise=0.02, permute=True)}
    mg = (CASES.get(name) or more[name])()
    levels = _levels(mg)
    monkeypatch.setenv("MGCFD_PLAN_DIGEST", "1")
    digests = {}
    for threads in ("1", "2", "3", "7"):
        monkeypatch.setenv("MGCFD_PLAN_THREADS", threads)
        try:
            rep = mgcfd.plan_audit(levels, mg.mesh_variant)
        except mgcfd.MgcfdError as e:               # (with the digest in it the report is never empty: the call says "not clean")
            rep = str(e)
        lines = [ln for ln in rep.splitlines() if ln.startswith("digest level") or "digest level" in ln]
        assert len(lines) == len(levels) and not [ln for ln in rep.splitlines() if ln.strip() and "digest level" not in ln and "plan audit" not in ln and not ln.startswith("level ")], rep
        digests[threads] = [ln.split("digest level")[1] for ln in lines]
    assert digests["1"] == digests["2"] == digests["3"] == digests["7"], digests
