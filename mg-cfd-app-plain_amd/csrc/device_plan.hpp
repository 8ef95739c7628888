// device_plan.hpp — plain structs shared by the host solver and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

#include "preprocess.hpp"

namespace mgcfd {

// Per-node arrays are stored structure-of-arrays with a common padded stride
// (stride = 256 * n_tiles): q[f*stride + i], f = 0..4 = the reference's `variables[i*5 + f]`;
// likewise old_variables, fluxes, residuals.
constexpr int kNumStateFields = 5;
static_assert(sizeof(EdgeW) == 32, "EdgeW must be 32 bytes");
static_assert(sizeof(ProlongW) == 24, "ProlongW must be 24 bytes");

// ff_variable + ff_flux_contribution_* (src/Base/globals.h:11-15), passed by value.
struct FarField {
    double var[5];
    double fc_mx[3], fc_my[3], fc_mz[3], fc_de[3];
};

// Where a halo "push" stores (kernels.hip: k_halo_push): the state buffers of up to kMaxPushPeers neighbouring ranks
// (peer access over xGMI, or the same device), the peers' segments of the message being consecutive slot ranges.
constexpr int kMaxPushPeers = 8;
struct PushPeers {
    double *base[kMaxPushPeers] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // peer p's state buffer [5][stride[p]]
    int64_t stride[kMaxPushPeers] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t first[kMaxPushPeers + 1] = {0, 0, 0, 0, 0, 0, 0, 0, 0};     // slots [first[p], first[p+1]) go to peer p
    int n = 0;
};

// The flags a push raises in its peers once its stores are on their way (ranks in different processes: kernels.hip,
// k_halo_push with flags, k_flags_wait): flag[p] = the word in peer p's memory, value = this message's sequence number.
struct PushFlags {
    unsigned long long *flag[kMaxPushPeers] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    unsigned long long value = 0;
    int n = 0;
};
// The message of a stage sent by the stage launch itself (k_flux_tile<..., PUSH>): the launch covers a rank's boundary tiles
// FIRST (workgroups [0, n_boundary)) and its interior tiles behind them; a boundary tile's epilogue stores the nodes the
// neighbours need (per-node lists: send_ptr / send_peer / send_target) into their ghost slots, and the boundary tile that
// finishes last raises the neighbours' flags — while the interior tiles of the same launch are still running.
struct StagePush {
    const int32_t *send_ptr = nullptr;      // [nel + 1] (library numbering): a node's entries
    const int8_t *send_peer = nullptr;      // [entries] position of the destination in peers
    const int32_t *send_target = nullptr;   // [entries] the node's index in that peer's numbering
    PushPeers peers;
    PushFlags flags;
    unsigned *ticket = nullptr;
    int32_t n_boundary = 0;
};

// A rank's flag words are rows of four (one per Runge-Kutta stage + one for the time-step all-reduce), one row per SOURCE rank.
constexpr int kMaxIpcRanks = 16;
struct FlagRows { int row[kMaxIpcRanks] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; int n = 0; };
// The all-reduce(MIN) of the time step between processes without a collective library: every rank stores its minimum into
// slot [parity][its rank] of every other rank's array and raises that rank's flag (k_min_publish); whoever has seen all flags
// holds all minima.
struct MinPublish {
    double *mins[kMaxIpcRanks] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};              // rank p's array [2][kMaxIpcRanks] (own rank: the local one)
    unsigned long long *flag[kMaxIpcRanks] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // rank p's flag word [this rank][3]
    int world = 0, me = 0, parity = 0;
    unsigned long long value = 0;
};

// Arguments of the time_step half of a fused flux + time_step launch (kernels.hip: k_flux_tile<FUSE>).
struct FusedStep {
    double rk_div = 1.0;                      // double(RK+1-j)
    double *step_factors = nullptr;
    const double *old_variables = nullptr;
    double *q_out = nullptr;                  // new variables (must differ from the launch's input state)
    const double *partial_min = nullptr;      // != nullptr on the first stage: finish compute_step_factor here
    int n_partial = 0;
    const double *volumes = nullptr;
    double *residuals = nullptr;              // != nullptr on the last stage: residuals = variables - old
    const int32_t *old_of_new = nullptr;
    unsigned long long *err = nullptr;
    int check = 0;
    // last stage extras
    // role 5 (a middle stage that applies the FIRST stage's time_step while it stages its input): the first stage's
    // fluxes and its divisor RK+1-0; the input state is then old_variables + (min_dt/volume/vin_div) * vin_flux
    const double *vin_flux = nullptr;
    double vin_div = 4.0;
    int check_vin = 0;                        // ... and the check sequence number of that absorbed time_step (earlier than `check`)
    double *sumsq_partial = nullptr;          // != nullptr (last stage): per-tile sums of squares of the residuals (calc_rms, validation.cpp:91-105)
    // look-ahead for the NEXT sweep on this level, from the state this launch produces:
    double *next_partial_min = nullptr;       // first half of compute_step_factor: per-tile minima of 0.5*cbrt(vol)/(|v|+c)
    const double *cbrt_vol = nullptr;
    double *next_legacy_sf = nullptr;         // mesh_name = fvcorr: next sweep's step factors 0.5/(sqrt(vol)*(|v|+c)) (cfd_loops.cpp:37-61) go here
    // a launch over PART of the level's tiles (a partitioned level: the tiles next to ghost nodes first, so that their
    // results can travel while the others are computed): tile_list[k] = tile of workgroup k; nullptr = all tiles
    const int32_t *tile_list = nullptr;
    int32_t n_list = 0;
    // > 0: the launch treats only nodes [0, nel_active) as present (a partitioned level whose ghosts are numbered last:
    // the ghost slots are then written by halo messages only — a peer may be storing into them during this launch)
    int64_t nel_active = 0;
};

// "Add up these partial sums" as an argument: k_sum_partials does only that; k_restrict can take it along.
struct SumTask {
    const double *partial = nullptr;      // [n]
    int n = 0;
    double *out = nullptr;                // the total
    double *ring = nullptr;               // nullptr, or: also append the total to the rms history
    int *count = nullptr;
    int cap = 0;
};

// Long rows (preprocess.hpp: LevelPlan::tail_*): device arrays of the entries the per-node loop leaves to the workgroup.
struct TailPlan {
    const int32_t *rows_main = nullptr;   // [n_slices] internal rows the per-node loop walks
    const int32_t *tile_ptr = nullptr;    // [n_tiles+1]
    const double2 *rec = nullptr;         // [total][3]: fx fy | fz k | (owner slot | code << 16) -
    const int32_t *begin = nullptr, *count = nullptr;   // [stride] a node's entries
    double2 *flux = nullptr;              // scratch [total][3]: the entry's five flux terms (48-byte records)
};

// Device pointers of one level's gather plan.
struct DevicePlan {
    int64_t nel = 0;
    int64_t stride = 0;                 // 256 * n_tiles
    int32_t n_slices = 0;
    int32_t *slice_row0 = nullptr, *rows_int = nullptr, *rows_bnd = nullptr, *nbr = nullptr;
    double *w = nullptr;                // [row][4 components][64 lanes]: signed, halved edge weights fx,fy,fz and k = -|e|*0.2f*0.5
    int32_t n_tiles = 0;
    int32_t pad_row = 0;                // index of a row of padding after the last row (nbr16, w, gat16)
    int32_t pad_chunk = 0;              // index of a chunk of padding after the last edge chunk (te_slots, te_w)
    uint16_t *nbr16 = nullptr;          // [row][64 lanes] tile-local codes
    int32_t *tile_halo = nullptr;        // [n_tiles][kHaloStride] ids of the staged halo nodes, -1 padded
    int32_t *tile_ovf_ptr = nullptr, *tile_ovf = nullptr;
    // edge-once tiles (preprocess.hpp: LevelPlan::te_*); edge_once == 0: not available on this level
    int vin_ok = 0;                     // no tile has overflow nodes or more than 256 halo nodes: role-5 launches allowed
    int lds_complete = 0;               // every halo node of every tile is staged in LDS (no overflow table entries)
    int32_t halo_max = 0;               // the largest halo of a tile (k_flux_free: how many workgroups may share a CU)
    int edge_once = 0;
    int32_t *te_chunk_ptr = nullptr, *te_count = nullptr;
    uint16_t *te_slots = nullptr, *gat16 = nullptr;
    double *te_w = nullptr;
    double *te_w3 = nullptr;            // [chunk*256 + p][3]: the same a-side weights as 24-byte records (k_flux_tile WMODE 2)
    // half rows (preprocess.hpp: LevelPlan::hr_*); half == 0: not available on this level
    int half = 0;
    int free_rows = 0;                  // the half-row plan is present at all (k_flux_free; `half`: k_flux_half can run on it too)
    int32_t hr_max_rows = 0;            // the most half rows a slice holds
    int free_wide = 0;                  // some tile's halo exceeds kHaloStride: k_flux_free stages from free_halo ([n_tiles][kFreeHaloStride])
    int32_t *free_halo = nullptr;
    int32_t hr_pad_row = 0;             // index of a half row of padding after the last one
    int32_t *hr_row0 = nullptr;         // [n_slices+1]
    uint32_t *hr_code = nullptr;        // [half row][64]
    uint16_t *hg16 = nullptr;
    double *hr_w = nullptr;             // [half row][3][64]
    int has_tail = 0;                   // long rows: some tile leaves entries to its workgroup (k_flux_tile<..., TAIL>)
    TailPlan tail;
    // two-phase ("fission") design point: per-edge arrays, edge-flux scratch [5][n_edges_pad], rows' edge references
    int64_t n_edges = 0, n_edges_pad = 0;
    int32_t *fe_ab = nullptr, *row_edge = nullptr;
    double *fe_w = nullptr, *edge_flux = nullptr;
    int32_t *old_of_new = nullptr;
    // transfer to/from the next-coarser level
    int32_t *child_ptr = nullptr, *child = nullptr;
    int32_t *child4 = nullptr;          // [nel_coarse][4] the first four children of every coarse node, -1 padded
    double *pro_w = nullptr;            // [row][w_own | w_other][64 lanes]
    int32_t *pro_p = nullptr;           // [row][64 lanes] coarse NEW id of the other end's parent
    int pro_tiled = 0;                  // k_prolong_tile can run (coarse residuals staged in LDS)
    int32_t *pro_tile_n = nullptr, *pro_tile_ids = nullptr;
    uint16_t *pro_s16 = nullptr, *pro_own16 = nullptr;
    int32_t *pro_parent = nullptr;
    double *pro_wsum = nullptr;
};

} // namespace mgcfd
