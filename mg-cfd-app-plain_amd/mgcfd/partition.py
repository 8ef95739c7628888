"""Partition one mesh level over ranks: owned nodes, ghost nodes, local edge lists, halo lists.

Host-side preparation for the multi-GPU "within a level" mode (BASELINE config 5).
Every rank owns a set of nodes and computes the fluxes of its own nodes only ("owner computes"):
it therefore needs every edge with at least one owned end point and a read-only GHOST copy of the
other end point when that belongs to another rank.  After each time_step the owners send the new
variables of those nodes to the ranks that hold ghosts of them — one message per neighbouring pair
per RK stage, 40 B per halo node.

Local numbering: owned nodes first (ascending global id), then ghosts (ascending global id).
Local edges keep the relative order they have in the global list inside each class, so every
owned node's incident edges are summed in the reference's order and the partitioned run is
bit-identical to the unpartitioned one.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np

from .meshgen import EDGE_DTYPE


@dataclass
class LevelPart:
    rank: int
    n_owned: int
    global_ids: np.ndarray                 # [n_local] global id of every local node (owned first, then ghosts)
    level: dict                            # read_grid()-shaped dict of the local part (for Solver.from_arrays)
    send: Dict[int, np.ndarray] = field(default_factory=dict)   # peer -> LOCAL ids of owned nodes the peer holds as ghosts
    recv: Dict[int, np.ndarray] = field(default_factory=dict)   # peer -> LOCAL ids of ghosts owned by the peer

    @property
    def n_local(self) -> int:
        return len(self.global_ids)


def slab_partition(coords: np.ndarray, n_parts: int, axis: int = 0) -> np.ndarray:
    """Equal-count slabs along one coordinate axis (good enough for box-like meshes)."""
    order = np.argsort(coords[:, axis], kind="stable")
    part = np.empty(len(coords), dtype=np.int64)
    part[order] = (np.arange(len(coords), dtype=np.int64) * n_parts) // len(coords)
    return part


def rcb_partition(coords: np.ndarray, n_parts: int) -> np.ndarray:
    """Recursive coordinate bisection: split the node set along the longest axis of its bounding box into two
    halves whose sizes are proportional to the parts each will hold, recurse.  Compact parts (a cube into 8 gives
    2 x 2 x 2 blocks: about a third of the halo of 8 slabs), equal counts to within a node, any n_parts >= 1."""
    coords = np.asarray(coords, dtype=np.float64)
    part = np.zeros(len(coords), dtype=np.int64)

    def split(ids: np.ndarray, first: int, count: int):
        if count == 1 or len(ids) == 0:
            part[ids] = first
            return
        left = count // 2
        box = coords[ids]
        axis = int(np.argmax(box.max(axis=0) - box.min(axis=0)))
        order = ids[np.argsort(box[:, axis], kind="stable")]
        cut = (len(ids) * left) // count
        split(order[:cut], first, left)
        split(order[cut:], first + left, count - left)

    split(np.arange(len(coords), dtype=np.int64), 0, int(n_parts))
    return part


def halo_volume(level: dict, part: np.ndarray) -> int:
    """Number of (node, neighbouring part) ghost copies a partition needs = messages' total length in nodes per exchange."""
    ni = int(level["n_internal"])
    a, b = level["edges"]["a"][:ni], level["edges"]["b"][:ni]
    pa, pb = part[a], part[b]
    cut = pa != pb
    # node b is a ghost on part pa, node a is a ghost on part pb (unique pairs)
    pairs = np.concatenate([np.stack([b[cut], pa[cut]], 1), np.stack([a[cut], pb[cut]], 1)])
    return len(np.unique(pairs, axis=0))


def partition_level(level: dict, part: np.ndarray) -> List[LevelPart]:
    """Split a read_grid()-shaped level dict (nel, volumes, coords, edges, n_internal, n_boundary,
    n_wall) by the node->rank vector `part`."""
    nel = int(level["nel"])
    edges = level["edges"]
    ni, nb, nw = int(level["n_internal"]), int(level["n_boundary"]), int(level["n_wall"])
    ea, eb = edges["a"], edges["b"]
    n_parts = int(part.max()) + 1
    internal = np.arange(ni)
    bnd = np.arange(ni, ni + nb)
    wall = np.arange(ni + nb, ni + nb + nw)
    owner_a = part[ea[internal]]
    owner_b = part[eb[internal]]
    parts: List[LevelPart] = []
    for r in range(n_parts):
        owned = np.flatnonzero(part == r)
        keep_int = internal[(owner_a == r) | (owner_b == r)]            # global order preserved
        keep_bnd = bnd[part[eb[bnd]] == r]
        keep_wall = wall[part[eb[wall]] == r]
        touched = np.union1d(ea[keep_int], eb[keep_int])
        ghosts = np.setdiff1d(touched, owned)
        gids = np.concatenate([owned, ghosts])
        local_of_global = np.full(nel, -1, dtype=np.int64)
        local_of_global[gids] = np.arange(len(gids))
        keep = np.concatenate([keep_int, keep_bnd, keep_wall])
        le = np.empty(len(keep), dtype=EDGE_DTYPE)
        le[:] = edges[keep]
        inner = np.arange(len(keep_int))
        le["a"][inner] = local_of_global[ea[keep_int]]
        le["b"] = local_of_global[eb[keep]]
        coords = None if level.get("coords") is None else np.ascontiguousarray(level["coords"])[gids]
        lvl = {"nel": len(gids), "volumes": np.ascontiguousarray(level["volumes"])[gids], "coords": coords,
               "edges": le, "n_internal": len(keep_int), "n_boundary": len(keep_bnd), "n_wall": len(keep_wall),
               "mg_map": None}
        parts.append(LevelPart(rank=r, n_owned=len(owned), global_ids=gids, level=lvl))
    # halo lists: ghosts of rank r owned by s, ascending global id on both sides
    for r, P in enumerate(parts):
        ghost_g = P.global_ids[P.n_owned:]
        ghost_owner = part[ghost_g]
        for s in np.unique(ghost_owner):
            g = ghost_g[ghost_owner == s]
            P.recv[int(s)] = (P.n_owned + np.flatnonzero(ghost_owner == s)).astype(np.int64)
            Q = parts[int(s)]
            local_in_s = np.searchsorted(Q.global_ids[:Q.n_owned], g)
            assert np.array_equal(Q.global_ids[local_in_s], g)
            Q.send[r] = local_in_s.astype(np.int64)
    return parts


# ----------------------------------------------------------------------------------------------
# A whole multigrid hierarchy partitioned over ranks
# ----------------------------------------------------------------------------------------------
@dataclass
class HierarchyPart:
    """One rank's share of a hierarchy: per level a LevelPart (owned nodes first, then ghosts, both ascending global id;
    level dict with a LOCAL mg_map) — what Solver.from_arrays(levels, n_owned=..., order_keys=...) takes."""
    rank: int
    levels: List[LevelPart]

    def solver_args(self):
        return ([p.level for p in self.levels], [p.n_owned for p in self.levels], [p.global_ids for p in self.levels])


def partition_hierarchy(levels: List[dict], part0: np.ndarray) -> List[HierarchyPart]:
    """Split every level of a hierarchy.  Level 0 follows `part0`; a coarse node goes to the rank that owns its first
    child (a childless one to rank 0).  Besides the flux ghosts (the other end of every edge with an owned end) a
    rank holds, per level, what the transfers need: the children of its owned coarse nodes (mgcfd_restrict computes a
    coarse node where it is owned) and the parent of every local fine node (mgcfd_prolong reads the parents of a fine
    node's neighbours; the local map must be total).  Local edge lists keep the global order, children are summed by
    global id (order_keys), so the partitioned V-cycle reproduces the whole mesh bit for bit."""
    n_levels = len(levels)
    n_parts = int(part0.max()) + 1
    owner = [np.asarray(part0, dtype=np.int64)]
    for l in range(n_levels - 1):
        m = np.asarray(levels[l]["mg_map"], dtype=np.int64)
        nc = int(levels[l + 1]["nel"])
        first_child = np.full(nc, len(m), dtype=np.int64)
        np.minimum.at(first_child, m, np.arange(len(m), dtype=np.int64))
        o = np.zeros(nc, dtype=np.int64)
        has = first_child < len(m)
        o[has] = owner[l][first_child[has]]
        owner.append(o)

    out = []
    for r in range(n_parts):
        local = []                                     # per level: sorted global ids of the local nodes
        owned = [np.flatnonzero(owner[l] == r) for l in range(n_levels)]
        for l in range(n_levels):
            L = levels[l]
            ni = int(L["n_internal"])
            a, b = L["edges"]["a"][:ni], L["edges"]["b"][:ni]
            mine = owner[l] == r
            touch = mine[a] | mine[b]
            need = [owned[l], a[touch], b[touch]]
            if l + 1 < n_levels:                        # children of owned coarse nodes
                m = np.asarray(L["mg_map"], dtype=np.int64)
                need.append(np.flatnonzero(owner[l + 1][m] == r))
            if l > 0:                                   # parents of every local node of the finer level
                need.append(np.asarray(levels[l - 1]["mg_map"], dtype=np.int64)[local[l - 1]])
            local.append(np.unique(np.concatenate(need)))
        parts = []
        for l in range(n_levels):
            L = levels[l]
            nel = int(L["nel"])
            ghosts = np.setdiff1d(local[l], owned[l])
            gids = np.concatenate([owned[l], ghosts])
            log = np.full(nel, -1, dtype=np.int64)
            log[gids] = np.arange(len(gids))
            ni, nb, nw = int(L["n_internal"]), int(L["n_boundary"]), int(L["n_wall"])
            ea, eb = L["edges"]["a"], L["edges"]["b"]
            mine = owner[l] == r
            internal = np.arange(ni)
            keep_int = internal[mine[ea[:ni]] | mine[eb[:ni]]]
            bnd = np.arange(ni, ni + nb)
            wall = np.arange(ni + nb, ni + nb + nw)
            keep_bnd = bnd[mine[eb[bnd]]]
            keep_wall = wall[mine[eb[wall]]]
            keep = np.concatenate([keep_int, keep_bnd, keep_wall])
            le = np.empty(len(keep), dtype=EDGE_DTYPE)
            le[:] = L["edges"][keep]
            le["a"][:len(keep_int)] = log[ea[keep_int]]
            le["b"] = log[eb[keep]]
            assert (le["a"][:len(keep_int)] >= 0).all() and (le["b"] >= 0).all()
            lvl = {"nel": len(gids), "volumes": np.ascontiguousarray(L["volumes"])[gids],
                   "coords": None if L.get("coords") is None else np.ascontiguousarray(L["coords"])[gids],
                   "edges": le, "n_internal": len(keep_int), "n_boundary": len(keep_bnd), "n_wall": len(keep_wall), "mg_map": None}
            parts.append(LevelPart(rank=r, n_owned=len(owned[l]), global_ids=gids, level=lvl))
        for l in range(n_levels - 1):                    # local maps: parent of every local fine node, in coarse local ids
            logc = np.full(int(levels[l + 1]["nel"]), -1, dtype=np.int64)
            logc[parts[l + 1].global_ids] = np.arange(parts[l + 1].n_local)
            lm = logc[np.asarray(levels[l]["mg_map"], dtype=np.int64)[parts[l].global_ids]]
            assert (lm >= 0).all()
            parts[l].level["mg_map"] = lm
        out.append(HierarchyPart(rank=r, levels=parts))
    # halo lists per level: every ghost is received from its owner, ascending global id on both sides
    for l in range(n_levels):
        for H in out:
            P = H.levels[l]
            ghost_g = P.global_ids[P.n_owned:]
            ghost_owner = owner[l][ghost_g]
            for s in np.unique(ghost_owner):
                g = ghost_g[ghost_owner == s]
                P.recv[int(s)] = (P.n_owned + np.flatnonzero(ghost_owner == s)).astype(np.int64)
                Q = out[int(s)].levels[l]
                pos = np.searchsorted(Q.global_ids[:Q.n_owned], g)
                assert np.array_equal(Q.global_ids[pos], g)
                Q.send[H.rank] = pos.astype(np.int64)
    return out
