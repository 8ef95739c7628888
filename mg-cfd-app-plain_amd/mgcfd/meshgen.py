"""Synthetic mesh generator that emits the reference's exact input formats.

The MG-CFD dataset release (fvcorr.domn.097K, Onera M6) is not shipped with the
reference repository, so every configuration is exercised on deterministic
synthetic meshes written in the reference's own text formats:

* ``input.dat``      key=value header + ``[levels]`` / ``[mg_mapping]`` sections
                     (reference reader: src/Base/io_enhanced.cpp:407-579)
* ``<mesh>``         ``nel number_of_edges`` then per node ``volume degree`` followed by
                     ``degree`` x (``neighbour wx wy wz``); neighbour >= 0 is a node id,
                     -1 a solid-wall face, -2 a far-field face (src/Base/io.cpp:56-137)
* ``<mesh>.coords``  ``x y z`` per node (src/Base/io.cpp:77-81)
* MG map file        ``mgc`` then ``mgc`` coarse indices (src/Base/io_enhanced.cpp:629-650)

Geometry: an n^3 lattice on the unit cube, optionally with a spherical cavity
(solid-wall faces), jittered coordinates, perturbed face areas, randomly
permuted node ids and shuffled neighbour lists so that nothing about the
numbering is structured.  Coarser levels are smaller lattices on the same cube;
fine->coarse maps are nearest-coarse-node.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

EDGE_DTYPE = np.dtype([("a", "<i8"), ("b", "<i8"), ("x", "<f8"), ("y", "<f8"), ("z", "<f8")])
"""Same 40-byte layout as the reference's ``edge_neighbour`` (src/Base/definitions.h:83)."""

MESH_FVCORR = 0
MESH_M6_WING = 2
MESH_LA_CASCADE = 3
MESH_ROTOR_37 = 4
MESH_CODES = {"fvcorr": MESH_FVCORR, "m6wing": MESH_M6_WING,
              "la_cascade": MESH_LA_CASCADE, "rotor37": MESH_ROTOR_37}


@dataclass
class LevelMesh:
    """One multigrid level in *file* form (what the mesh text file lists)."""
    nel: int
    volumes: np.ndarray            # [nel] f8
    coords: np.ndarray             # [nel, 3] f8
    nbr_ptr: np.ndarray            # [nel+1] i8  CSR over the per-node neighbour lists
    nbr_idx: np.ndarray            # [nnz] i8    neighbour id, -1 (solid wall) or -2 (far field)
    nbr_w: np.ndarray              # [nnz, 3] f8 area-weighted normal as written in the file
    mg_map: Optional[np.ndarray] = None   # [nel] i8 fine -> coarse (next level), None on the last

    @property
    def number_of_edges(self) -> int:
        node = np.repeat(np.arange(self.nel, dtype=np.int64), np.diff(self.nbr_ptr))
        return int(np.count_nonzero(self.nbr_idx < node))


@dataclass
class MultigridMesh:
    mesh_name: str
    levels: List[LevelMesh] = field(default_factory=list)
    size: int = 1

    @property
    def mesh_variant(self) -> int:
        return MESH_CODES[self.mesh_name]


_DIRS = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.int64)


def make_box_level(n: int, *, seed: int = 0, cavity_radius: float = 0.0, jitter: float = 0.0,
                   area_noise: float = 0.0, volume_noise: float = 0.0, permute: bool = True,
                   shuffle_neighbours: bool = True) -> LevelMesh:
    """n^3 lattice on [0,1]^3.  Outer faces are far-field (-2); nodes inside a sphere
    of ``cavity_radius`` around the cube centre are removed and the faces that looked at
    them become solid wall (-1)."""
    assert n >= 2
    rng = np.random.default_rng(seed)
    h = 1.0 / (n - 1)
    ii, jj, kk = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    ijk = np.stack([ii.ravel(), jj.ravel(), kk.ravel()], axis=1).astype(np.int64)
    lattice_xyz = ijk / float(n - 1)
    if cavity_radius > 0.0:
        keep = np.linalg.norm(lattice_xyz - 0.5, axis=1) > cavity_radius
        if keep.all():   # radius smaller than the spacing: always drop the node nearest the centre
            keep[np.argmin(np.linalg.norm(lattice_xyz - 0.5, axis=1))] = False
    else:
        keep = np.ones(len(ijk), dtype=bool)
    lat_id = np.full(n * n * n, -1, dtype=np.int64)
    nel = int(keep.sum())
    lat_id[keep] = np.arange(nel)
    ijk_k = ijk[keep]

    # half-width factors: a dual cell is halved along each axis on which the node sits on the hull
    half = np.where((ijk_k == 0) | (ijk_k == n - 1), 0.5, 1.0)
    volumes = (h ** 3) * half.prod(axis=1)
    if volume_noise > 0.0:
        volumes = volumes * (1.0 + volume_noise * rng.uniform(-1.0, 1.0, nel))

    src = np.repeat(np.arange(nel, dtype=np.int64), 6)
    d = np.tile(_DIRS, (nel, 1))
    nb_ijk = np.repeat(ijk_k, 6, axis=0) + d
    inside = ((nb_ijk >= 0) & (nb_ijk < n)).all(axis=1)
    nb_lin = np.where(inside, (nb_ijk[:, 0].clip(0, n - 1) * n + nb_ijk[:, 1].clip(0, n - 1)) * n
                      + nb_ijk[:, 2].clip(0, n - 1), 0)
    nb = np.where(inside, lat_id[nb_lin], -2)           # outside the cube: far field
    nb = np.where(inside & (nb < 0), -1, nb)            # removed node: solid wall
    # face area: h^2 times the half factors of the two transverse axes
    axis = np.abs(d).argmax(axis=1)
    half6 = np.repeat(half, 6, axis=0)
    transverse = np.ones(len(src))
    for ax in range(3):
        transverse *= np.where(axis == ax, 1.0, half6[:, ax])
    area = (h ** 2) * transverse
    if area_noise > 0.0:
        # symmetric per-edge perturbation so both listings of an internal edge agree
        lo = np.minimum(src, np.where(nb >= 0, nb, src))
        hi = np.maximum(src, np.where(nb >= 0, nb, src))
        key = (lo * 1000003 + hi * 7919 + axis * 13) % 2147483647
        area = area * (1.0 + area_noise * np.sin(key.astype(np.float64) * 12.9898 + seed))
    w = d.astype(np.float64) * area[:, None]

    xyz = lattice_xyz[keep].copy()
    if jitter > 0.0:
        interior = ((ijk_k > 0) & (ijk_k < n - 1)).all(axis=1)
        xyz[interior] += jitter * h * rng.uniform(-1.0, 1.0, (int(interior.sum()), 3))

    if permute:
        perm = rng.permutation(nel).astype(np.int64)      # new id of old node i
    else:
        perm = np.arange(nel, dtype=np.int64)
    new_src = perm[src]
    new_nb = np.where(nb >= 0, perm[np.clip(nb, 0, nel - 1)], nb)
    tie = rng.random(len(src)) if shuffle_neighbours else np.arange(len(src), dtype=np.float64)
    order = np.lexsort((tie, new_src))
    counts = np.bincount(new_src, minlength=nel)
    ptr = np.zeros(nel + 1, dtype=np.int64)
    np.cumsum(counts, out=ptr[1:])
    inv = np.empty(nel, dtype=np.int64)
    inv[perm] = np.arange(nel)
    return LevelMesh(nel=nel, volumes=volumes[inv].copy(), coords=xyz[inv].copy(), nbr_ptr=ptr,
                     nbr_idx=new_nb[order].copy(), nbr_w=w[order].copy())


# ---------------------------------------------------------------------------------------
# A level of MIXED element types on one lattice of points: what an M6-like mesh is made of
# ---------------------------------------------------------------------------------------
_DIAG_FACE = np.array([[1, 1, 0], [-1, -1, 0], [0, 1, 1], [0, -1, -1], [1, 0, 1], [-1, 0, -1]], dtype=np.int64)
_DIAG_BODY = np.array([[1, 1, 1], [-1, -1, -1]], dtype=np.int64)


def make_mixed_level(n: int, *, seed: int = 0, prism_layers: int = 3, tet_shell: int = 2, jitter: float = 0.0,
                     area_noise: float = 0.0, volume_noise: float = 0.0, permute: bool = True,
                     shuffle_neighbours: bool = True) -> LevelMesh:
    """n^3 points with the connectivity of three element types (the Onera-M6 release is hexahedra-dominant with prism layers on
    the wing and tetrahedra towards the far field; the files are not shipped):

    * a HEXAHEDRAL core: the lattice's six neighbours (interior degree 6);
    * PRISM layers on the wall z = 0 (a solid wall, -1): the ``prism_layers`` point layers next to it carry the in-plane
      diagonal of a triangulated wall surface extruded upwards (degree 8);
    * a TETRAHEDRAL far field: the ``tet_shell`` point layers next to the five other faces (far field, -2) are connected as
      the Kuhn triangulation of their cells — three face diagonals and the body diagonal (degree up to 14).

    A diagonal exists only between two nodes of the same region, and every direction comes with its opposite at equal
    weight, so the dual faces of an interior node close (a uniform state stays uniform).  Internal degrees 3 ... 14 (18 %
    of the nodes not 6); at n = 67: 300,763 nodes / 1,004,901 internal edges (3.3 per node; the M6: 300 K / 930 K)."""
    assert n >= 4
    rng = np.random.default_rng(seed)
    h = 1.0 / (n - 1)
    ii, jj, kk = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    ijk = np.stack([ii.ravel(), jj.ravel(), kk.ravel()], axis=1).astype(np.int64)
    nel = len(ijk)
    prism = ijk[:, 2] < prism_layers
    near_far = (np.minimum(ijk[:, 0], n - 1 - ijk[:, 0]) < tet_shell) | (np.minimum(ijk[:, 1], n - 1 - ijk[:, 1]) < tet_shell) | \
               (n - 1 - ijk[:, 2] < tet_shell)
    tet = near_far & ~prism
    dirs = np.concatenate([_DIRS, _DIAG_FACE, _DIAG_BODY])                    # 6 axis | 6 face diagonals | 2 body diagonals
    nd = len(dirs)
    half = np.where((ijk == 0) | (ijk == n - 1), 0.5, 1.0)
    volumes = (h ** 3) * half.prod(axis=1)
    if volume_noise > 0.0:
        volumes = volumes * (1.0 + volume_noise * rng.uniform(-1.0, 1.0, nel))
    src = np.repeat(np.arange(nel, dtype=np.int64), nd)
    d = np.tile(dirs, (nel, 1))
    kind = np.tile(np.arange(nd), nel)                                        # which direction
    nb_ijk = np.repeat(ijk, nd, axis=0) + d
    inside = ((nb_ijk >= 0) & (nb_ijk < n)).all(axis=1)
    nb = np.where(inside, (nb_ijk[:, 0].clip(0, n - 1) * n + nb_ijk[:, 1].clip(0, n - 1)) * n + nb_ijk[:, 2].clip(0, n - 1), 0)
    axis_dir = kind < 6
    # an axis direction that leaves the cube is a boundary face: the wall below, the far field everywhere else
    code = np.where(inside, nb, np.where(d[:, 2] < 0, -1, -2))
    # diagonals: both ends in the prism region (only the in-plane pair), or both ends in the tetrahedral region (all eight)
    both_prism = prism[src] & prism[nb] & inside
    both_tet = tet[src] & tet[nb] & inside
    in_plane = (kind == 6) | (kind == 7)
    keep = axis_dir | (in_plane & both_prism) | ((kind >= 6) & both_tet)
    keep &= axis_dir | inside                                                 # (a diagonal that leaves the cube is no face)
    src, d, kind, code, inside = src[keep], d[keep], kind[keep], code[keep], inside[keep]
    # dual-face areas: axis faces as the box's (halved on the hull); diagonal faces a fixed share of h^2
    half_s = half[src]
    transverse = np.ones(len(src))
    ax = np.abs(d).argmax(axis=1)
    for a in range(3):
        transverse *= np.where((kind < 6) & (ax != a), half_s[:, a], 1.0)
    length = np.linalg.norm(d, axis=1)
    area = (h ** 2) * np.where(kind < 6, transverse, np.where(kind < 12, 0.30, 0.18))
    if area_noise > 0.0:
        other = np.where(code >= 0, code, src)
        lo, hi = np.minimum(src, other), np.maximum(src, other)
        key = (lo * 1000003 + hi * 7919 + (kind // 2) * 13) % 2147483647
        area = area * (1.0 + area_noise * np.sin(key.astype(np.float64) * 12.9898 + seed))
    w = d.astype(np.float64) / length[:, None] * area[:, None]
    xyz = ijk / float(n - 1)
    if jitter > 0.0:
        interior = ((ijk > 0) & (ijk < n - 1)).all(axis=1)
        xyz[interior] += jitter * h * rng.uniform(-1.0, 1.0, (int(interior.sum()), 3))
    perm = rng.permutation(nel).astype(np.int64) if permute else np.arange(nel, dtype=np.int64)
    new_src = perm[src]
    new_nb = np.where(code >= 0, perm[np.clip(code, 0, nel - 1)], code)
    tie = rng.random(len(src)) if shuffle_neighbours else np.arange(len(src), dtype=np.float64)
    order = np.lexsort((tie, new_src))
    counts = np.bincount(new_src, minlength=nel)
    ptr = np.zeros(nel + 1, dtype=np.int64)
    np.cumsum(counts, out=ptr[1:])
    inv = np.empty(nel, dtype=np.int64)
    inv[perm] = np.arange(nel)
    return LevelMesh(nel=nel, volumes=volumes[inv].copy(), coords=xyz[inv].copy(), nbr_ptr=ptr,
                     nbr_idx=new_nb[order].copy(), nbr_w=w[order].copy())


def make_mixed_multigrid(sizes: Sequence[int], mesh_name: str = "m6wing", *, seed: int = 0, jitter: float = 0.0,
                         area_noise: float = 0.0, volume_noise: float = 0.0, permute: bool = True, **mixed) -> MultigridMesh:
    """A hierarchy whose level 0 is a mixed-element level (make_mixed_level) over plain lattice levels, nearest-node maps."""
    mg = MultigridMesh(mesh_name=mesh_name)
    for l, n in enumerate(sizes):
        if l == 0:
            mg.levels.append(make_mixed_level(n, seed=seed, jitter=jitter, area_noise=area_noise, volume_noise=volume_noise, permute=permute, **mixed))
        else:
            mg.levels.append(make_box_level(n, seed=seed + 101 * l, area_noise=area_noise, volume_noise=volume_noise, permute=permute))
    for l in range(len(sizes) - 1):
        mg.levels[l].mg_map = nearest_map(mg.levels[l], mg.levels[l + 1])
    return mg


def make_random_graph_level(nel: int, *, degree: int = 6, seed: int = 0, boundary_fraction: float = 0.1) -> LevelMesh:
    """A deliberately NON-geometric level: every node is joined to `degree` random others, so no
    numbering has locality and any 256-node cluster touches far more outside nodes than an LDS
    tile holds.  Exercises the overflow path of the flux kernel and the robustness of the
    renumbering; physically meaningless (random weights, far-field faces on some nodes)."""
    rng = np.random.default_rng(seed)
    a = np.repeat(np.arange(nel, dtype=np.int64), degree // 2)
    b = rng.integers(0, nel, len(a))
    keep = a != b
    a, b = a[keep], b[keep]
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    pairs = np.unique(np.stack([lo, hi], axis=1), axis=0)
    lo, hi = pairs[:, 0], pairs[:, 1]
    w = rng.normal(size=(len(lo), 3)) * 1e-3
    n_bnd = int(boundary_fraction * nel)
    bnodes = rng.choice(nel, n_bnd, replace=False).astype(np.int64)
    bcode = np.where(rng.random(n_bnd) < 0.3, -1, -2).astype(np.int64)
    bw = rng.normal(size=(n_bnd, 3)) * 1e-3
    src = np.concatenate([lo, hi, bnodes])
    nb = np.concatenate([hi, lo, bcode])
    ww = np.concatenate([w, -w, bw])
    order = np.lexsort((rng.random(len(src)), src))
    counts = np.bincount(src, minlength=nel)
    ptr = np.zeros(nel + 1, dtype=np.int64)
    np.cumsum(counts, out=ptr[1:])
    return LevelMesh(nel=nel, volumes=rng.uniform(0.5e-6, 2e-6, nel), coords=rng.random((nel, 3)), nbr_ptr=ptr,
                     nbr_idx=nb[order].copy(), nbr_w=ww[order].copy())


_TET_EDGES = ((0, 1, 2, 3), (0, 2, 1, 3), (0, 3, 1, 2), (1, 2, 0, 3), (1, 3, 0, 2), (2, 3, 0, 1))   # (i, j, the other two)


def make_tet_level(nel: int, *, seed: int = 0, wall_below: float = 0.9) -> LevelMesh:
    """A genuinely unstructured level: the Delaunay tetrahedralisation of ``nel`` random points in the unit
    cube with its MEDIAN-DUAL finite-volume metrics (the kind of mesh the reference's datasets hold: node
    degrees from 5 to 50+, about 7.7 edges per node, no ordering locality).  Per internal edge the area vector
    is the sum, over the tetrahedra around it, of the dual quadrilateral (edge midpoint, face centroid, cell
    centroid, face centroid); node volume = a quarter of every incident tetrahedron; a hull node gets one
    boundary face that closes its dual cell exactly (solid wall where that face looks down, -z, far field
    elsewhere), so a uniform state is preserved to rounding."""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    xyz = rng.random((nel, 3))
    tri = Delaunay(xyz)
    tets = tri.simplices.astype(np.int64)
    p = xyz[tets]                                               # [T, 4, 3]
    cen = p.mean(axis=1)
    vol6 = np.abs(np.einsum("ij,ij->i", np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), p[:, 3] - p[:, 0]))
    volumes = np.bincount(tets.ravel(), weights=np.repeat(vol6 / 24.0, 4), minlength=nel)
    ea, eb, es = [], [], []
    for i, j, k, l in _TET_EDGES:
        mid = 0.5 * (p[:, i] + p[:, j])
        fk = (p[:, i] + p[:, j] + p[:, k]) / 3.0
        fl = (p[:, i] + p[:, j] + p[:, l]) / 3.0
        s = 0.5 * np.cross(cen - mid, fl - fk)
        s *= np.sign(np.einsum("ij,ij->i", s, p[:, j] - p[:, i]))[:, None]     # from i towards j
        a, b = tets[:, i], tets[:, j]
        swap = a > b
        ea.append(np.where(swap, b, a)); eb.append(np.where(swap, a, b)); es.append(np.where(swap[:, None], -s, s))
    ea, eb, es = np.concatenate(ea), np.concatenate(eb), np.concatenate(es)
    key = ea * nel + eb
    uniq, inv = np.unique(key, return_inverse=True)
    w = np.stack([np.bincount(inv, weights=es[:, c], minlength=len(uniq)) for c in range(3)], axis=1)
    lo, hi = uniq // nel, uniq % nel                            # w points from lo to hi
    # what is missing from each hull node's closed surface is its boundary face
    out = np.zeros((nel, 3))
    for c in range(3):
        out[:, c] = np.bincount(lo, weights=w[:, c], minlength=nel) - np.bincount(hi, weights=w[:, c], minlength=nel)
    hull = np.unique(tri.convex_hull).astype(np.int64)
    bw = -out[hull]
    bcode = np.where(-bw[:, 2] > wall_below * np.linalg.norm(bw, axis=1), -1, -2).astype(np.int64)
    src = np.concatenate([lo, hi, hull])
    nb = np.concatenate([hi, lo, bcode])
    ww = np.concatenate([w, -w, bw])
    order = np.lexsort((rng.random(len(src)), src))
    counts = np.bincount(src, minlength=nel)
    ptr = np.zeros(nel + 1, dtype=np.int64)
    np.cumsum(counts, out=ptr[1:])
    return LevelMesh(nel=nel, volumes=volumes, coords=xyz, nbr_ptr=ptr, nbr_idx=nb[order].copy(), nbr_w=ww[order].copy())


def make_tet_multigrid(sizes: Sequence[int], mesh_name: str = "m6wing", *, seed: int = 0) -> MultigridMesh:
    """Hierarchy of independent Delaunay levels (``sizes`` = node counts), nearest-coarse-node maps."""
    mg = MultigridMesh(mesh_name=mesh_name)
    for l, n in enumerate(sizes):
        mg.levels.append(make_tet_level(n, seed=seed + 101 * l))
    for l in range(len(sizes) - 1):
        mg.levels[l].mg_map = nearest_map(mg.levels[l], mg.levels[l + 1])
    return mg


def make_hub_level(spokes: int, *, scale: float = 1e-4, seed: int = 0) -> LevelMesh:
    """One node joined to ``spokes`` others (a row far longer than a tile is wide), one far-field face on the hub,
    random weights of size ``scale``: non-physical on purpose — with ``mesh_name = fvcorr`` (undamped) and weights
    above ~1e-4 the state goes negative after a few iterations, which is what the error-path tests need."""
    rng = np.random.default_rng(seed)
    nel = spokes + 1
    w = rng.normal(size=(spokes, 3)) * scale
    bw = rng.normal(size=(1, 3)) * scale
    src = np.concatenate([np.zeros(spokes, dtype=np.int64), np.arange(1, nel, dtype=np.int64), [0]])
    nb = np.concatenate([np.arange(1, nel, dtype=np.int64), np.zeros(spokes, dtype=np.int64), [-2]])
    ww = np.concatenate([w, -w, bw])
    order = np.lexsort((np.arange(len(src)), src))
    counts = np.bincount(src, minlength=nel)
    ptr = np.zeros(nel + 1, dtype=np.int64)
    np.cumsum(counts, out=ptr[1:])
    return LevelMesh(nel=nel, volumes=rng.uniform(1e-6, 2e-6, nel), coords=rng.random((nel, 3)), nbr_ptr=ptr,
                     nbr_idx=nb[order].copy(), nbr_w=ww[order].copy())


def nearest_map(fine: LevelMesh, coarse: LevelMesh) -> np.ndarray:
    from scipy.spatial import cKDTree
    _, idx = cKDTree(coarse.coords).query(fine.coords, k=1)
    return idx.astype(np.int64)


def make_multigrid(sizes: Sequence[int], mesh_name: str = "m6wing", *, seed: int = 0,
                   cavity_radius: float = 0.0, jitter: float = 0.0, area_noise: float = 0.0,
                   volume_noise: float = 0.0, permute: bool = True) -> MultigridMesh:
    """Lattice hierarchy, e.g. ``sizes=(67, 55, 48, 43)`` for the M6-like 4-level case
    (300,763 / 166,375 / 110,592 / 79,507 nodes)."""
    mg = MultigridMesh(mesh_name=mesh_name)
    for l, n in enumerate(sizes):
        mg.levels.append(make_box_level(n, seed=seed + 101 * l, cavity_radius=cavity_radius,
                                        jitter=jitter if l == 0 else 0.0, area_noise=area_noise,
                                        volume_noise=volume_noise, permute=permute))
    for l in range(len(sizes) - 1):
        mg.levels[l].mg_map = nearest_map(mg.levels[l], mg.levels[l + 1])
    return mg


# ---------------------------------------------------------------------------------------
# In-memory equivalent of the reference's mesh reader (for benches that skip the text files)
# ---------------------------------------------------------------------------------------
def to_edge_arrays(level: LevelMesh, mesh_variant: int):
    """Edge list exactly as the reference's ``read_grid`` would build it from the text file
    (src/Base/io.cpp:84-177): an entry is recorded when ``neighbour < node`` as
    ``{a=neighbour, b=node, w}``; w is negated for internal edges (all edges if fvcorr);
    output order [internal | boundary(-1) | wall(-2)], file order within each class.
    Returns ``(edges[EDGE_DTYPE], n_internal, n_boundary, n_wall)``."""
    node = np.repeat(np.arange(level.nel, dtype=np.int64), np.diff(level.nbr_ptr))
    rec = level.nbr_idx < node
    a = level.nbr_idx[rec]
    b = node[rec]
    w = level.nbr_w[rec].copy()
    flip = np.ones(len(a), dtype=bool) if mesh_variant == MESH_FVCORR else (a >= 0)
    w[flip] *= -1.0
    cls = np.where(a >= 0, 0, np.where(a == -1, 1, 2))
    order = np.argsort(cls, kind="stable")
    edges = np.empty(len(a), dtype=EDGE_DTYPE)
    edges["a"] = a[order]
    edges["b"] = b[order]
    edges["x"] = w[order, 0]
    edges["y"] = w[order, 1]
    edges["z"] = w[order, 2]
    return edges, int((cls == 0).sum()), int((cls == 1).sum()), int((cls == 2).sum())


# ---------------------------------------------------------------------------------------
# Writers
# ---------------------------------------------------------------------------------------
def _fmt(x: float) -> str:
    return repr(float(x))     # shortest round-trip decimal: parses back to the same double


def write_level(level: LevelMesh, path: str) -> None:
    with open(path, "w") as f:
        f.write(f"{level.nel} {level.number_of_edges}\n")
        ptr, idx, w, vol = level.nbr_ptr, level.nbr_idx, level.nbr_w, level.volumes
        for i in range(level.nel):
            s, e = int(ptr[i]), int(ptr[i + 1])
            parts = [f"{_fmt(vol[i])} {e - s}"]
            for k in range(s, e):
                parts.append(f"{int(idx[k])} {_fmt(w[k, 0])} {_fmt(w[k, 1])} {_fmt(w[k, 2])}")
            f.write(" ".join(parts))
            f.write("\n")
    with open(path + ".coords", "w") as f:
        for x, y, z in level.coords:
            f.write(f"{_fmt(x)} {_fmt(y)} {_fmt(z)}\n")


def write_mg_map(mapping: np.ndarray, path: str) -> None:
    with open(path, "w") as f:
        f.write(f"{len(mapping)}\n")
        f.write("\n".join(str(int(v)) for v in mapping))
        f.write("\n")


def write_input(mg: MultigridMesh, directory: str, *, stem: str = "mesh", dat_name: str = "input.dat") -> str:
    """Write all files of a multigrid input into ``directory``; returns the input.dat path."""
    os.makedirs(directory, exist_ok=True)
    nl = len(mg.levels)
    lines = ["# synthetic MG-CFD input (mgcfd.meshgen)", f"size = {mg.size}", f"num_levels = {nl}",
             f"mesh_name = {mg.mesh_name}", "", "[levels]"]
    for l, lvl in enumerate(mg.levels):
        name = f"{stem}.L{l}.dat"
        write_level(lvl, os.path.join(directory, name))
        lines.append(f"{l} = {name}")
    if nl > 1:
        lines += ["", "[mg_mapping]"]
        for l in range(nl - 1):
            name = f"{stem}.mg_L{l}_to_L{l + 1}.dat"
            write_mg_map(mg.levels[l].mg_map, os.path.join(directory, name))
            lines.append(f"{l} = {name}")
    dat = os.path.join(directory, dat_name)
    with open(dat, "w") as f:
        f.write("\n".join(lines) + "\n")
    return dat
