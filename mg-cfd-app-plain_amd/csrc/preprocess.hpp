// preprocess.hpp — host-side construction of the device gather structures.
//
// The reference walks an edge list and scatter-adds into both end points
// (src/Kernels/flux_loops.cpp:133-136).  On the GPU every node instead GATHERS
// its incident edges: nodes are renumbered for locality, grouped into slices of
// 64 consecutive nodes (one wavefront) and each slice stores its incidence lists
// as a sliced-ELLPACK block, row r holding the r-th incident edge of all 64
// nodes contiguously.  The r-th entries are ordered by ORIGINAL edge index
// (internal, then boundary, then wall), which is exactly the order in which the
// reference's serial loops add into fluxes[node] — so a per-node sequential sum
// reproduces the reference's floating-point result bit for bit, with no atomics.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "mgcfd.h"

namespace mgcfd {

constexpr int kSlice = 64;   // wavefront width
constexpr int kTile = 256;   // nodes per tile = one 256-thread workgroup = 4 slices
// node records a tile can stage in LDS.  A CU's 160 KiB of LDS is handed out in granules (1,280 B as measured: a kernel that asks
// for 53,760 B runs three workgroups per CU, one that asks for 53,792 runs TWO), and the stage kernels keep a few words beside
// the records (per-wave minima and sums): 559 * 96 B = 53,664 B leaves them 96 B inside the 42nd granule.  With 560 records the
// first and the last stage of every sweep ran at two workgroups per CU: 20.6 / 21.1 us against 18.3 / 19.1 (profiles/README.md).
constexpr int kTileCap = 559;
// edge-once tiles: a tile's internal edges are evaluated in chunks of one edge per thread; the
// fluxes of up to kMaxEdgeChunks chunks wait in registers, then replace the node records in LDS
// (kMaxEdgeChunks * 256 * 40 B must fit kTileCap * 96 B)
constexpr int kHaloStride = kTileCap - kTile;   // halo ids a tile can stage (device table: fixed stride, -1 padded)
static_assert(kTileCap >= 2 * kTile, "every thread has a halo slot (the kernels stage one unconditionally)");
static_assert(kHaloStride <= 2 * kTile, "the staging code reads at most two halo ids per thread");
// prolongation tiles: distinct coarse nodes (own parents + neighbours' parents) a fine tile may stage in LDS (40 B each)
constexpr int kProCap = 768;
constexpr int kEdgeChunk = 256;
constexpr int kMaxEdgeChunks = 5;
constexpr uint32_t kHalfForeign = 1u << 24;
constexpr uint32_t kHalfMirror = 1u << 25;    // the edge lies inside the tile and this is its only evaluation: the other end takes the
                                              // negated terms (k_flux_free adds them to that node's LDS sum; k_flux_half hands them over by position)
constexpr uint32_t kHalfPad = 0x00007FFFu;   // (low 15 bits = kT16Pad)
constexpr int kHalfMaxRows = 5;            // half rows (edges a node evaluates) a thread keeps the results of in registers
constexpr int kFreeHaloStride = 2 * kTile; // halo ids a tile of the order-free kernel can stage from its OWN table (two per thread: 56-byte LDS images,
                                           // 768 of them + the sums = 53,248 B, three workgroups per CU) where some tile's halo exceeds kHaloStride
constexpr int kFreeCap4Halo = 290;         // the largest halo with which four workgroups of the order-free kernel share a CU (kernels.hip: kFreeCap4 - kTile)
constexpr int kFreeMaxRows = 32;           // ... and what a slice may hold at all: the order-free kernel walks the rows beyond five in a loop
constexpr int kHalfTileRows = 21;          // half rows of a tile's four slices together (21 * 64 * 40 B = the whole LDS tile)
constexpr int kHalfSlots = kHalfTileRows * kSlice;   // flux-term slots of a tile in LDS: 5 fields x 1344 x 8 B = 52.5 KiB (its own array in k_flux_half)
constexpr int kHalfLdsD2 = (5 * kHalfSlots + 1) / 2;   // double2 the half-row kernel's flux terms take in LDS (they replace the node records)
static_assert(kMaxEdgeChunks * kEdgeChunk * 40 <= kTileCap * 96, "edge fluxes must fit the LDS tile");

// neighbour codes in Sell::nbr
constexpr int32_t kRoleB = 1 << 30;     // set when THIS node is the edge's 'b' end (else it is 'a')
constexpr int32_t kIdMask = kRoleB - 1;
// 16-bit tile-local codes (LevelPlan::nbr16): bit 15 = role, low 15 bits = LDS slot
// (own nodes 0..255, staged halo 256..kTileCap-1), kTileCap+k = k-th entry of the tile's
// overflow table (halo beyond the LDS capacity), or one of the specials below.
constexpr uint32_t kT16RoleB = 0x8000u;
constexpr uint32_t kT16SlotMask = 0x7FFFu;
constexpr uint32_t kT16Pad = 0x7FFFu;
constexpr uint32_t kT16Wall = 0x7FFEu;    // solid wall face   (reference neighbour code -1)
constexpr uint32_t kT16Far = 0x7FFDu;     // far-field face    (reference neighbour code -2)
constexpr int32_t kCodeWall = -1;       // reference neighbour code -1: solid wall   ("boundary" edges)
constexpr int32_t kCodeFar = -2;        // reference neighbour code -2: far field    ("wall" edges)
constexpr int32_t kCodePad = -3;

struct EdgeW { double x, y, z, k; };    // 32 B, one per (node, incident edge)
struct ProlongW { double w_own, w_other; int32_t p_own, p_other; };   // 24 B

struct LevelPlan {
    int64_t nel = 0;
    int32_t n_slices = 0;
    std::vector<int32_t> new_of_old, old_of_new;   // node permutation

    // ---- flux gather (sliced ELL) ----
    std::vector<int32_t> slice_row0;   // [n_slices+1] first row of each slice; entry index = row*64 + lane
    std::vector<int32_t> rows_int;     // [n_slices] rows holding internal edges
    std::vector<int32_t> rows_bnd;     // [n_slices] rows holding boundary/far-field faces (after the internal rows)
    std::vector<int32_t> nbr;          // [rows*64] neighbour (new id | role bit) or a kCode*
    // ---- tiles: 256 consecutive nodes staged in LDS together with their halo ----
    int32_t n_tiles = 0;
    std::vector<int32_t> tile_halo_ptr;   // [n_tiles+1]
    std::vector<int32_t> tile_halo;       // global new ids of the halo nodes a tile stages (ascending, <= kTileCap-kTile each)
    std::vector<int32_t> tile_ovf_ptr;    // [n_tiles+1]
    std::vector<int32_t> tile_ovf;        // global new ids of halo nodes that did not fit the LDS tile
    std::vector<uint16_t> nbr16;          // [rows*64] 16-bit tile-local codes (see kT16*)
    double halo_mean = 0.0; int32_t halo_max = 0; int64_t halo_overflow_refs = 0, halo_total = 0;
    bool ordered_by_boxes = false;        // node order from rcb_split (coordinate boxes) instead of cluster_order
    bool ghosts_last = false;             // partitioned level: new ids [0, n_owned) are the owned nodes, the ghosts follow
    std::vector<EdgeW> w;              // [rows*64]
    //   internal, this node = a:  (x,y,z) = -0.5*e   k = -|e|*smoothing*0.5   (flux_kernel.elemfunc.c:130-140)
    //   internal, this node = b:  (x,y,z) = +0.5*e   k = same
    //   wall (-1):                (x,y,z) = e                                  (flux_boundary_kernel.elemfunc.c:37-45)
    //   far field (-2):           (x,y,z) = 0.5*e                              (flux_wall_kernel.elemfunc.c:51-53)
    // ---- edge-once tiles: every internal edge that touches a tile, listed once per tile in
    //      ORIGINAL edge order; entry p of tile t is evaluated by thread p % 256 in chunk p / 256 ----
    bool edge_once = false;               // false: some tile holds more than kMaxEdgeChunks*256 edges (node gather only)
    int32_t te_max = 0; double te_mean = 0.0;
    std::vector<int32_t> te_chunk_ptr;    // [n_tiles+1] first chunk of each tile
    std::vector<int32_t> te_count;        // [n_tiles] edges of each tile
    std::vector<uint16_t> te_slots;       // [chunk][2][256] LDS slot (as nbr16, no role bit) of end points a and b; kT16Pad = no edge
    std::vector<double> te_w;             // [chunk][4][256] a-side weights: -0.5*e (x,y,z) and k = -|e|*smoothing*0.5
    std::vector<uint16_t> gat16;          // [rows*64] internal rows: position p of the entry's edge in its tile's list
                                          //   | kT16RoleB when this node is the edge's b end (it gets -F); kT16Pad
    // ---- half rows (k_flux_half): every internal edge that touches a tile is EVALUATED by exactly one of its end points
    //      in the tile (an edge inside the tile by the less loaded of the two, an edge cut by the tile boundary by the end
    //      inside), so a tile streams one 28-byte row entry per edge instead of one of 26-34 bytes per end point, and
    //      evaluates each edge once.  The evaluator leaves the five flux terms in LDS at position (half row within the tile) * 64 + lane;
    //      every node then adds its incident edges in row order from there (hg16: position | kT16RoleB when the entry's
    //      edge was evaluated by the OTHER end point: that node adds the negated terms, which is what the reference's
    //      expressions for the other end evaluate to bit for bit) ----
    bool free_rows = false;               // the half-row plan exists (k_flux_free can run): every tile's halo is staged in LDS
    bool half = false;                    // ... and k_flux_half can run on it too: no slice needs more than kHalfMaxRows rows, no tile more than
                                          // kHalfTileRows, no long rows
    int32_t hr_max_rows = 0, hr_max_tile_rows = 0;
    bool free_wide = false;               // some tile's halo exceeds kHaloStride: the order-free kernel stages from free_halo, and the half rows' slots
                                          // count positions in THAT list (256 + position among ALL the tile's halo ids, ascending)
    std::vector<int32_t> free_halo;       // [n_tiles][kFreeHaloStride] (free_wide only), -1 padded
    std::vector<int32_t> hr_row0;         // [n_slices+1] first half row of each slice
    std::vector<uint32_t> hr_code;        // [half rows*64] low 16 bits as nbr16 (the other end's LDS slot | kT16RoleB when the OWNING end
                                          //   is the edge's b end), bits 16-23 the owning node's thread, kHalfForeign when the slot
                                          //   sits in another node's lane (that lane reads the owner's record from LDS too), kHalfMirror when
                                          //   the other end lies in the tile and leaves the evaluation to this entry; kHalfPad
    std::vector<double> hr_w;             // [half rows][3][64] the evaluator's weights (as w.x, w.y, w.z of its own entry)
    std::vector<uint16_t> hg16;           // [rows*64] internal rows: where the entry's flux terms are | kT16RoleB = subtract; kT16Pad
    int64_t hr_entries = 0, hr_padding = 0, hr_foreign = 0;
    // ---- long rows: a node's internal entries beyond its tile's row limit.  The per-node loop stops at the limit;
    //      the whole workgroup then evaluates the remaining entries one per thread, leaves the five results in a
    //      global scratch, and every owner adds its own in row order — the summation order is unchanged, but the
    //      workgroup no longer waits for its highest-degree wave (tetrahedral meshes, hubs) ----
    bool has_tail = false;
    std::vector<int32_t> rows_main;       // [n_slices] internal rows the per-node loop walks (<= rows_int)
    std::vector<int32_t> tail_tile_ptr;   // [n_tiles+1] a tile's range in the tail arrays (multiples of 16 entries)
    std::vector<double> tail_rec;         // [tail_total][6]: the entry's weights fx, fy, fz, k (as w), then one word
                                          //   holding (tile-local slot of the owning node) | (neighbour code as nbr16) << 16
                                          //   — owner kT16Pad = padding —, then padding to 48 bytes (three 16-byte loads)
    std::vector<int32_t> tail_begin, tail_count;   // [nel] a node's entries: contiguous, in row order
    int64_t tail_total = 0;
    // ---- two-phase ("fission") design point: edge fluxes to memory, then a node-centred sum ----
    std::vector<int32_t> fe_ab;           // [2][n_internal] end points (new ids) of every internal edge, original order
    std::vector<double> fe_w;             // [4][n_internal] a-side weights -0.5*e (x,y,z) and k
    std::vector<int32_t> row_edge;        // [rows*64] internal rows: the entry's edge (index into fe_*) | bit 31 when this
                                          //   node is its b end; -1 = padding / boundary rows
    int64_t n_internal_entries = 0;    // = 2 * internal edges
    double pad_fraction = 0.0;         // padding / useful entries in the internal rows
    int64_t pad_entries = 0;           // padding entries in the internal rows

    // ---- restriction to the next-coarser level: coarse-centred CSR of children ----
    std::vector<int32_t> child_ptr;    // [nel_coarse+1]
    std::vector<int32_t> child;        // fine NEW ids, ascending ORIGINAL fine id (mg_loops.cpp:119-142 order)

    // ---- prolongation from the next-coarser level (shares slice_row0/rows_int) ----
    std::vector<ProlongW> pro;         // [rows*64] only the internal rows are meaningful
    std::vector<int32_t> pro_parent;   // [nel] coarse NEW id of the node's parent; ~id when the node coincides with it
    std::vector<double> pro_wsum;      // [nel] sum of weights in reference order (1.0 when coincident)
    // tiled form: the coarse nodes a fine tile refers to, staged in LDS by k_prolong_tile
    bool pro_tiled = false;            // false: some tile refers to more than kProCap coarse nodes (k_prolong gathers from HBM)
    std::vector<int32_t> pro_tile_n;   // [n_tiles] distinct coarse nodes of the tile
    std::vector<int32_t> pro_tile_ids; // [n_tiles][kProCap] their coarse NEW ids, ascending, -1 padded
    std::vector<uint16_t> pro_s16;     // [rows*64] position of the entry's other-end parent in that list
    std::vector<uint16_t> pro_own16;   // [nel] position of the node's own parent
};

// n_owned < nel marks the nodes with ORIGINAL id >= n_owned as ghosts of a partitioned level:
// read-only copies of nodes another rank owns.  They are staged and gathered like any node but
// get no incidence rows of their own (their flux stays zero, time_step leaves them unchanged, and
// the halo exchange overwrites them).
struct PlanOptions {
    // 0: keep the caller's numbering; 1: breadth-first (Cuthill-McKee) bands; 2: compact clusters of
    // kTile nodes grown greedily over the mesh graph (smallest halo per tile; default)
    int ordering = 2;
    bool degree_sort = true;           // inside each tile, sort nodes by degree (less ELL padding per slice)
    int64_t n_owned = -1;              // -1: every node is owned
    bool long_rows = true;             // cut rows at a per-tile limit and evaluate the rest workgroup-wide (LevelPlan::tail_*)
    int tile_order = 2;                // dispatch order of the tiles within an XCD's range (preprocess.cpp): 0 as clustered, 1 costliest first, 2 the cheapest last (one round's share at most) where the tiles' longest rows differ, 3 the same on every level
    int tile_curve = 1;                // the complete owned tiles along a space-filling curve through their centroids before that: 0 as clustered, 1 Morton, 2 Hilbert (needs coords)
    int threads = 0;                   // host threads for the per-tile part of the plan (0: up to 8, MGCFD_PLAN_THREADS overrides); the plan does not depend on it
};

// `edges` are the level's final edge weights (after adjust/dampen).  coarse_new_of_old is
// the permutation of the next-coarser level (needed to express parents in new ids), empty
// on the last level.
void build_level_plan(const mgcfd_level_desc &lvl, const std::vector<mgcfd_edge> &edges,
                      const PlanOptions &opt, LevelPlan &plan);
uint64_t plan_digest(const LevelPlan &plan);
// child_order_key (may be null): per fine node, the key by which a coarse node's children are summed (ascending);
// null = the node's own index, the reference's order.  A partitioned level passes global ids, so the mean over the
// children keeps the whole mesh's summation order whatever the local numbering.  n_owned_fine < nel: fine nodes with
// index >= n_owned_fine are ghosts (no incidence rows, hence no prolongation entries).
void build_transfer_plan(const mgcfd_level_desc &fine, const std::vector<mgcfd_edge> &fine_edges,
                         const double *coarse_coords, int64_t nel_coarse,
                         const std::vector<int32_t> &coarse_new_of_old, LevelPlan &fine_plan,
                         const int64_t *child_order_key = nullptr, int64_t n_owned_fine = -1);

// Every index the kernels form from a freshly built plan against the size of what it indexes ("" = all in range);
// nel_coarse: size of the next-coarser level when a transfer plan was built into P, else -1.
std::string audit_level_plan(const mgcfd_level_desc &lvl, const LevelPlan &plan, int64_t nel_coarse = -1);

// Edge-weight preconditioning exactly as the reference does before its loop
// (src/Kernels/validation.cpp:28-75, src/euler3d_cpu_double.cpp:337-352).
void adjust_and_dampen(const mgcfd_level_desc &lvl, int mesh_variant, std::vector<mgcfd_edge> &edges);

// ff_variable + ff_flux_contribution_* (src/Kernels/cfd_loops.h:85-119); out[17].
void far_field_constants(double *out17);

} // namespace mgcfd
