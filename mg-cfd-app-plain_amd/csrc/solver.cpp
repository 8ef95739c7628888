// solver.cpp — device-resident solver behind the C ABI of include/mgcfd.h.
//
// Owns, per multigrid level, the state arrays the reference's main() owns
// (src/euler3d_cpu_double.cpp:138-162) in HBM, in the renumbered node order of the
// level's gather plan, and exposes the reference's kernel set one call per loop plus the
// V-cycle state machine (src/euler3d_cpu_double.cpp:371-694).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <functional>
#include <future>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "kernels.hpp"
#include "mesh.hpp"
#include "mgcfd.h"
#include "preprocess.hpp"

namespace mgcfd {

static thread_local std::string g_last_error;

struct HipError : std::runtime_error {
    explicit HipError(const std::string &m) : std::runtime_error(m) {}
};

#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            throw HipError(std::string(#expr) + " failed: " + hipGetErrorString(_e));                \
    } while (0)

template <typename T> static T *dev_alloc(size_t n)
{
    void *p = nullptr;
    HIP_CHECK(hipMalloc(&p, (n ? n : 1) * sizeof(T)));
    return static_cast<T *>(p);
}

template <typename T> static T *dev_upload(const std::vector<T> &v)
{
    T *p = dev_alloc<T>(v.size());
    if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return p;
}

// A level's device arrays live in ONE block (a creation is one hipMalloc and a destruction one hipFree per level instead
// of 55 of each).  build_solver first lists them — which member points where, from which host array it is filled — with
// no call into the runtime (so that the listing and the repacking into device layouts can run while the device is still
// waking up), then allocates the block, sets the members and copies.
struct LevelStaging {
    struct Item { std::function<void(char *)> place; const void *src; size_t bytes; size_t offset; };
    std::vector<Item> items;
    std::vector<std::shared_ptr<void>> keep;      // arrays repacked for the device, alive until they have been copied
    size_t total = 0;
    static constexpr size_t kAlign = 4096;
    template <typename F> void alloc(F *&field, size_t bytes, const void *src = nullptr)
    {
        const size_t off = total;
        total += (std::max<size_t>(bytes, 1) + kAlign - 1) / kAlign * kAlign;
        F **where = &field;
        items.push_back({[where, off](char *base) { *where = reinterpret_cast<F *>(base + off); }, src, src ? bytes : 0, off});
    }
    template <typename F, typename T> void upload(F *&field, const std::vector<T> &v) { alloc(field, v.size() * sizeof(T), v.data()); }
    template <typename F, typename T> void upload(F *&field, std::vector<T> &&v)
    {
        auto held = std::make_shared<std::vector<T>>(std::move(v));
        keep.push_back(held);
        upload(field, *held);
    }
};

struct EventPair { hipEvent_t start, stop; int level, loop; bool is_flux_internal; int launches; };

// The halo exchange of one partitioned level as the C++ host runs it (mgcfd_rank_* / mgcfd_group_*): ONE packed message
// per exchange holding every peer's segment (one pack and one unpack launch whatever the number of peers), one buffer
// set per Runge-Kutta stage (a set is reused a whole sweep later), and the level's tiles split into those next to ghost
// nodes ("boundary": they read ghosts, and every node a peer needs lies in one of them) and the rest ("interior"),
// which run while the message travels.
struct HaloExchange {
    static constexpr int kSets = 3;
    std::vector<int> peer;                               // neighbouring ranks, ascending
    std::vector<int64_t> send_off, recv_off;             // [n_peers+1] node offsets of the peers' segments
    int32_t *send_idx = nullptr, *recv_idx = nullptr;    // device: library node ids, all peers concatenated
    double *send_buf[kSets] = {nullptr, nullptr, nullptr}, *recv_buf[kSets] = {nullptr, nullptr, nullptr};   // device: [nodes][5]
    int32_t *tiles_boundary = nullptr, *tiles_interior = nullptr, *tiles_all = nullptr;   // (all = boundary then interior: one launch per stage)
    int32_t n_boundary = 0, n_interior = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t packed[kSets] = {nullptr, nullptr, nullptr}, arrived[kSets] = {nullptr, nullptr, nullptr};
    hipEvent_t reduced = nullptr, gathered = nullptr, joined = nullptr;   // the in-process all-reduce of the time step; graph capture joins
    double *gmin = nullptr;                              // device [world]: [0] = the group's minimum time step (in-process groups)
    // in-process groups: every rank reads the others' minima IN PLACE (k_min_over_peers), so a rank's minimum of sweep k must not
    // be overwritten by its reduction of sweep k + 1 while a rank that is no halo neighbour — nothing on the device orders the two —
    // may still be reading it: the minima alternate between two words by sweep parity (a reader two sweeps behind cannot exist:
    // the writer's gather of sweep k + 1 waits for the reader's reduction event of sweep k + 1, which lies behind its read of sweep k)
    double *min_par = nullptr;                           // device [2]: this rank's minimum of even / odd sweeps
    int min_parity = 0;                                  // ... and which of them the NEXT sweep writes
    const double **peer_scalars[2] = {nullptr, nullptr}; // device [world] per parity: where every rank of the group keeps that minimum
    // the sweep as a captured graph per buffer rotation (one host call per sweep instead of ~25): RCCL ranks
    hipGraphExec_t sweep_graph[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [rotation (+ 3 x parity of the minima: in-process groups)]
    int64_t graph_iters[6][MGCFD_NUM_LOOPS] = {{0}};
    bool graph_failed = false;
    int64_t sweeps_replayed = 0;          // ... launched from one of them so far (mgcfd_rank_graph_status)
    // in-process groups, direct mode: a rank's message is stored by ONE launch straight into its peers' ghost slots
    // (k_halo_push) — no message buffers, no second stream, no unpack; a stage's boundary tiles wait for the peers'
    // previous-stage events instead
    std::vector<int32_t> recv_idx_host;                  // recv_idx on the host (a peer builds its push targets from it)
    int32_t *push_target = nullptr;                      // device [total_send]: slot k's node in the numbering of the peer it goes to
    hipEvent_t bdone[3] = {nullptr, nullptr, nullptr};   // [stage] this rank's boundary tiles and push of that stage are enqueued up to here
    bool direct = false;
    // ranks in different PROCESSES, direct mode (mgcfd_rank_ipc_*): the peers' state buffers and flag words opened through HIP IPC
    bool ipc = false;
    unsigned long long *flags = nullptr;                 // device [kMaxIpcRanks][4]: word [r][slot] is raised by rank r (slot = stage; 3 = the time-step all-reduce)
    double *gmins = nullptr;                             // device [2][kMaxIpcRanks]: every rank's time-step minimum of this sweep (parity) / the last
    bool ipc_all = false;                                // every rank of the job is attached: the time-step all-reduce goes through the flags too
    unsigned long long sweep_seq = 0;                    // all-reduces published so far
    double *all_mins[kMaxIpcRanks] = {};                 // opened (own: gmins): rank r's gmins
    unsigned long long *all_flag[kMaxIpcRanks] = {};     // opened: rank r's flag word [this rank][3]
    unsigned *ticket = nullptr;                          // device: workgroups of the running push that are done
    int *ipc_timeouts = nullptr;                         // device: waits that gave up (a peer that never arrived)
    bool ipc_unacknowledged = false;                     // a synchronising call has found ipc_timeouts != 0 and nobody has asked mgcfd_rank_ipc_status since
    unsigned long long seq = 0;                          // messages pushed so far (all ranks push in lockstep)
    double *peer_state[kMaxPushPeers][3] = {};           // opened: peer k's three state buffers, as the peer numbers them
    int64_t peer_stride[kMaxPushPeers] = {};
    int peer_rot_delta[kMaxPushPeers] = {};              // (peer's rotation - ours) mod 3 when the buffers were exchanged
    unsigned long long *peer_flag[kMaxPushPeers] = {};   // opened: the words of peer k's flag array this rank raises
    std::vector<void *> ipc_opened;                      // every mapping opened (closed again in mgcfd_rank_detach / the destructor)
    int32_t *node_send_ptr = nullptr, *node_send_target = nullptr;   // device: the message per NODE (StagePush: a stage that sends it itself)
    int8_t *node_send_peer = nullptr;
    int64_t total_send() const { return send_off.empty() ? 0 : send_off.back(); }
    int64_t total_recv() const { return recv_off.empty() ? 0 : recv_off.back(); }
};

struct DeviceLevel {
    mgcfd_level_desc info{};             // sizes only (pointers nulled)
    std::vector<mgcfd_edge> edges;       // final edge weights, original order
    LevelPlan plan;                      // host copy (permutations for get/set)
    DevicePlan dp;
    // SoA state, stride = dp.stride: q = the reference's `variables`
    double *q = nullptr, *old_variables = nullptr, *fluxes = nullptr, *residuals = nullptr;   // [5][stride]
    double *q_alt = nullptr;             // [5][stride] second state buffer for the fused RK stages
    // The three state buffers change roles after every fused sweep (no copy<double>(old, variables)):
    // variables = state[rot], q_alt = state[rot+1], old_variables = state[rot+2] (mod 3).
    double *state[3] = {nullptr, nullptr, nullptr};
    int rot = 0;
    void apply_rot() { q = state[rot % 3]; q_alt = state[(rot + 1) % 3]; old_variables = state[(rot + 2) % 3]; }
    double *tile_sumsq = nullptr;        // [n_tiles] per-tile sums of squares of the residuals, written by a last stage on request
    bool want_sumsq = false;             // the next fused sweep's last stage also fills tile_sumsq (cycle driver, level 0)
    bool have_sumsq = false;             // ... and did
    int64_t row_bytes = 0;               // bytes of the incidence rows (ids + weights)
    bool min_ahead = false;              // the launch that produced the CURRENT variables looked ahead: partial_min holds the
                                         // first half of compute_step_factor for them (global time step), or sf_alt holds
                                         // their step factors (mesh_name = fvcorr, local time step)
    double *sf_alt = nullptr;            // [stride] fvcorr look-ahead target; swapped with step_factors when consumed
    int sf_par = 0;                      // parity of those swaps (part of the graph keys)
    double *sfb[2] = {nullptr, nullptr};
    void apply_sf() { step_factors = sfb[sf_par & 1]; sf_alt = sfb[(sf_par & 1) ^ 1]; }
    double *step_factors = nullptr, *volumes = nullptr, *cbrt_vol = nullptr;                  // [stride]
    double *min_dt = nullptr;            // global-min time step scalar (after the reduction)
    double *partial_min = nullptr;       // one partial minimum per step-factor workgroup
    double *sumsq = nullptr, *partials = nullptr;
    int n_partials = 0;
    bool fluxes_zero = true;             // fluxes[] is logically zero (the flux launch need not read it)
    bool fluxes_stale = false;           // ... but its memory has not been zeroed (lazy zero after a fused time_step)
    bool residuals_stale = false;        // residuals[] = variables - old_variables of the last sweep (validation.cpp:77-89), not written
                                         // yet: a single-level run's sweeps overwrite it unread; settle_residuals writes it on demand
    bool sweep_flux0_done = false;       // mgcfd_sweep_flux0 ran since mgcfd_sweep_begin
    int stage_next = 0;                  // mgcfd_sweep_stage: the stage expected next (0 = a sweep may start)
    double *stage_out = nullptr;         // ... and the buffer the last one wrote (MGCFD_ARR_STAGE)
    int64_t n_owned = 0;                 // < nel on a partitioned level: original ids >= n_owned are ghosts
    std::vector<std::pair<int32_t *, int64_t>> halo_plans;   // device id lists of the halo messages
    std::unique_ptr<HaloExchange> hx;    // the C++-side exchange of a partitioned level (mgcfd_rank_set_halo)
    bool has_transfer = false;           // plan to the next-coarser level present
    void *block = nullptr;               // the one allocation behind every array listed at creation (LevelStaging)
    size_t block_bytes = 0;
    bool in_block(const void *p) const { return block && p >= block && p < static_cast<const char *>(block) + block_bytes; }
    int64_t iters[MGCFD_NUM_LOOPS] = {0};
    double times[MGCFD_NUM_LOOPS] = {0};
    double flux_time = 0.0;
    int64_t flux_launches = 0;
    // MGCFD_OPT_TIMING = 4 (per-loop times by attribution): how often each loop ran on this level and how often under events
    int64_t att_sweeps = 0, att_sweeps_sampled = 0;      // smoothing sweeps (flux, compute_step, time_step, indirect_rw run together)
    int64_t att_calls[MGCFD_NUM_LOOPS] = {0}, att_calls_sampled[MGCFD_NUM_LOOPS] = {0};   // restrict (booked on the coarse level) and prolong
};

} // namespace mgcfd

using namespace mgcfd;

struct mgcfd_mesh { HostMesh mesh; };

struct mgcfd_solver {
    int device = 0;
    int mesh_variant = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::vector<DeviceLevel> L;
    FarField ff{};
    double ff17[17] = {0};
    unsigned long long *err = nullptr;       // device: packed (cell << 8 | code), ~0 = clean
    int opt_exact = 1, opt_timing = 0, opt_indirect_rw = 0, opt_check = 1, opt_variant = -1, opt_fuse = 1, opt_graph = 0;
    // (MGCFD_LAZY_RESIDUAL=0: every sweep's last stage writes residuals[] — for A/B measurements)
    int opt_rank_split = 1;
    bool opt_lazy_residual = !(std::getenv("MGCFD_LAZY_RESIDUAL") && std::atoi(std::getenv("MGCFD_LAZY_RESIDUAL")) == 0);
    // check_for_invalid_variables: every checked launch carries its sequence number since the host last read the
    // error word, so the EARLIEST failing time_step wins, as in the reference (kernels.hip: err_key)
    int check_seq = 0;
    int64_t invalid_cell = -1; int invalid_cycle = -1;                     // where the last reported invalid state was found
    int next_check() { if (!opt_check) return 0; if (check_seq < (1 << 22)) check_seq++; return check_seq; }
    int force_check = -1;                    // >= 0: the sequence number the next fused stage carries (a stage launched in two parts)
    int timing_stride = 8;                  // OPT_TIMING == 2: bracket the flux launches of every Nth sweep; == 4: every Nth sweep / transfer of a level runs unfused under events
    double att_total = 0.0;                 // OPT_TIMING == 4: GPU seconds of the cycle batches (one event pair around each)
    bool probe_first_stage_only = false;    // (a sampled sweep of OPT_TIMING == 4 runs the indirect_rw probe behind its first stage only)
    int64_t sweep_counter = 0;
    bool in_timed_group = false;
    struct SweepGraph { hipGraphExec_t exec = nullptr; int64_t iters[MGCFD_NUM_LOOPS] = {0}; bool ahead_after = false; int rot_after = 0; int sf_par_after = 0; bool sumsq_after = false, res_stale_after = false; };
    std::map<uint64_t, SweepGraph> sweep_graphs;   // captured smoothing sweeps, keyed by (level, options)
    struct CycleGraph { hipGraphExec_t exec = nullptr; std::vector<std::vector<int64_t>> iters; std::vector<bool> ahead_after, res_stale_after; std::vector<int> rot_after, sf_par_after; };
    std::map<uint64_t, CycleGraph> cycle_graphs;   // captured whole multigrid cycles, keyed by options
    static constexpr int kRmsRing = 4096;
    double *rms_ring = nullptr;                    // level-0 sum of squares of the cycles run since the last read-back
    int *rms_count = nullptr;
    std::vector<EventPair> pending;
    std::vector<hipEvent_t> free_events;

    ~mgcfd_solver();
    void use_device() const { HIP_CHECK(hipSetDevice(device)); }
    DeviceLevel &level(int l)
    {
        if (l < 0 || l >= static_cast<int>(L.size())) throw std::invalid_argument("level out of range");
        return L[static_cast<size_t>(l)];
    }

    // ---- timing --------------------------------------------------------------------------
    hipEvent_t get_event()
    {
        if (pending.size() >= 8192) fold_events();
        if (!free_events.empty()) { hipEvent_t e = free_events.back(); free_events.pop_back(); return e; }
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        return e;
    }
    void fold_events()
    {
        if (pending.empty()) return;
        HIP_CHECK(hipStreamSynchronize(stream));
        for (auto &p : pending) {
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, p.start, p.stop));
            DeviceLevel &lv = L[static_cast<size_t>(p.level)];
            lv.times[p.loop] += double(ms) * 1e-3;
            if (p.is_flux_internal) { lv.flux_time += double(ms) * 1e-3; lv.flux_launches += p.launches; }
            free_events.push_back(p.start);
            free_events.push_back(p.stop);
        }
        pending.clear();
    }
    struct Timed {
        mgcfd_solver *s; EventPair p; bool on;
        // launches > 1: one pair brackets that many back-to-back launches of the flux kernel, so
        // only the first pays the dispatch latency an event pair adds (the others' dispatch
        // overlaps the predecessor) and the mean agrees with rocprofv3's kernel duration
        Timed(mgcfd_solver *s_, int level, int loop, bool flux_internal = false, int launches = 1)
            : s(s_), on(!s_->in_timed_group && (s_->opt_timing == 1 || (s_->opt_timing == 2 && flux_internal)))
        {
            if (!on) return;
            p = EventPair{s->get_event(), s->get_event(), level, loop, flux_internal, launches};
            if (launches > 1) s->in_timed_group = true;
            HIP_CHECK(hipEventRecord(p.start, s->stream));
        }
        ~Timed()
        {
            if (!on) return;
            (void)hipEventRecord(p.stop, s->stream);
            s->pending.push_back(p);
            if (p.launches > 1) s->in_timed_group = false;
        }
    };

    // ---- operations ----------------------------------------------------------------------
    void op_copy_old(int l)
    {
        DeviceLevel &lv = level(l);
        settle_residuals(lv);
        HIP_CHECK(hipMemcpyAsync(lv.old_variables, lv.q, sizeof(double) * 5 * lv.dp.stride, hipMemcpyDeviceToDevice, stream));
    }
    // first half of compute_step_factor; reduce_to_scalar also folds the workgroups' partial
    // minima into the one fp64 scalar the kernel-granular API / the all-reduce hook work on
    void op_step_factor_local(int l, bool fuse_copy_old, bool reduce_to_scalar)
    {
        DeviceLevel &lv = level(l);
        lv.min_ahead = false;                       // partial_min is rewritten (same values if it was ahead)
        if (fuse_copy_old) settle_residuals(lv);
        double *old = fuse_copy_old ? lv.old_variables : nullptr;
        if (opt_exact) exact::launch_step_factor_local(stream, lv.info.nel, lv.dp.stride, lv.q, lv.cbrt_vol, lv.step_factors, lv.partial_min, old);
        else fast::launch_step_factor_local(stream, lv.info.nel, lv.dp.stride, lv.q, lv.cbrt_vol, lv.step_factors, lv.partial_min, old);
        if (reduce_to_scalar) exact::launch_min_reduce(stream, lv.info.nel, lv.partial_min, lv.min_dt);
    }
    void op_step_factor_apply(int l)
    {
        DeviceLevel &lv = level(l);
        if (opt_exact) exact::launch_step_factor_apply(stream, lv.info.nel, lv.min_dt, lv.volumes, lv.step_factors);
        else fast::launch_step_factor_apply(stream, lv.info.nel, lv.min_dt, lv.volumes, lv.step_factors);
    }
    // Write the residual of the last sweep if its last stage left it out (single-level runs: the next sweep would only
    // overwrite it).  Both operands are still in place — `variables` and the sweep's start state, which the buffer rotation
    // made old_variables — and the subtraction is the one the stage would have done, so the bits are the same.  Called by
    // everything that reads residuals[] and by everything that is about to change either operand.
    void settle_residuals(DeviceLevel &lv)
    {
        if (!lv.residuals_stale) return;
        lv.residuals_stale = false;
        if (opt_exact) exact::launch_residual(stream, lv.dp.stride, lv.old_variables, lv.q, lv.residuals);
        else fast::launch_residual(stream, lv.dp.stride, lv.old_variables, lv.q, lv.residuals);
    }
    // materialise a logically-zero flux array before anything reads its memory
    void settle_fluxes(DeviceLevel &lv)
    {
        if (!lv.fluxes_stale) return;
        HIP_CHECK(hipMemsetAsync(lv.fluxes, 0, sizeof(double) * 5 * lv.dp.stride, stream));
        lv.fluxes_stale = false;
    }
    // fused = true: also copy old_variables <- variables, and leave the "/ volume" half of the
    // global time step to the first time_step of the sweep (returns true in that case)
    bool op_step_factor(int l, bool fused = false, bool copy_old = true)
    {
        DeviceLevel &lv = level(l);
        Timed t(this, l, MGCFD_LOOP_COMPUTE_STEP);
        bool apply_pending = false;
        if (mesh_variant == MGCFD_MESH_FVCORR) {
            double *old = (fused && copy_old) ? lv.old_variables : nullptr;
            if (old) settle_residuals(lv);
            if (opt_exact) exact::launch_step_factor_legacy(stream, lv.info.nel, lv.dp.stride, lv.q, lv.volumes, lv.step_factors, old);
            else fast::launch_step_factor_legacy(stream, lv.info.nel, lv.dp.stride, lv.q, lv.volumes, lv.step_factors, old);
        } else {
            op_step_factor_local(l, fused && copy_old, !fused);
            if (fused) apply_pending = true;
            else op_step_factor_apply(l);
        }
        lv.iters[MGCFD_LOOP_COMPUTE_STEP] += lv.info.nel;
        return apply_pending;
    }
    // The parts of a level's plan that only a non-default option reaches are uploaded when an option first asks for them
    // (at creation: nothing but the 32-bit neighbour codes of levels whose indirect_rw probe runs the L1-gather form).  Called
    // at the end of creation and from mgcfd_set_option — never from a launch path, so never inside a stream capture.
    void upload_optional_plans(DeviceLevel &lv)
    {
        LevelPlan &P = lv.plan;
        const int v = opt_variant;
        auto take = [](auto &vec) { auto *p = dev_upload(vec); vec.clear(); vec.shrink_to_fit(); return p; };
        // order-free / half rows: MGCFD_OPT_EXACT = 0 (the automatic variant takes the order-free kernel), or bits 5 / 6
        if (lv.dp.free_rows && !lv.dp.hr_code && (!opt_exact || (v >= 0 && (v & (32 | 64))))) {
            lv.dp.hr_row0 = dev_upload(P.hr_row0);
            lv.dp.hr_code = take(P.hr_code);
            lv.dp.hr_w = take(P.hr_w);
            if (lv.dp.free_wide) lv.dp.free_halo = take(P.free_halo);
        }
        if (lv.dp.free_rows && !lv.dp.hg16 && v >= 0 && (v & 32)) lv.dp.hg16 = take(P.hg16);
        // edge-once tiles (bit 1) and indexed weights (bit 4)
        if (lv.dp.edge_once && !lv.dp.gat16 && v >= 0 && (v & (2 | 16))) {
            lv.dp.te_chunk_ptr = dev_upload(P.te_chunk_ptr);
            lv.dp.te_count = dev_upload(P.te_count);
            lv.dp.gat16 = take(P.gat16);
            {   // the a-side weights as 24-byte records, one per listed edge (k_flux_tile WMODE 2: indexed weights)
                const size_t n_chunks = P.te_w.size() / (4 * kEdgeChunk);
                std::vector<double> w3(n_chunks * kEdgeChunk * 3);
                for (size_t c = 0; c < n_chunks; c++)
                    for (size_t ln = 0; ln < size_t(kEdgeChunk); ln++)
                        for (size_t k = 0; k < 3; k++) w3[(c * kEdgeChunk + ln) * 3 + k] = P.te_w[(c * 4 + k) * kEdgeChunk + ln];
                lv.dp.te_w3 = dev_upload(w3);
            }
            lv.dp.te_slots = take(P.te_slots);
            lv.dp.te_w = take(P.te_w);
        }
        // the two-phase design point (bit 2)
        if (!lv.dp.fe_ab && v >= 0 && (v & 4)) {
            lv.dp.fe_ab = take(P.fe_ab);
            lv.dp.fe_w = take(P.fe_w);
            lv.dp.row_edge = take(P.row_edge);
        }
        // 32-bit neighbour codes: the L1-gather form of the indirect_rw probe (bit 3, or a level the tile form cannot run on)
        if (!lv.dp.nbr && ((v >= 0 && (v & 8)) || !lv.dp.lds_complete || lv.dp.has_tail)) lv.dp.nbr = take(P.nbr);
    }
    void upload_optional_plans()
    {
        use_device();
        for (DeviceLevel &lv : L) upload_optional_plans(lv);
    }
    // MGCFD_OPT_FLUX_VARIANT = -1 (auto): the edge-length factor is recomputed from the weights (8 of 34 bytes per entry
    // less to stream, one sqrt more per entry).  The fused stages run equally fast either way while a level's rows fit the
    // Infinity Cache (bench level: 20.15 against 20.2 us) and 13-17 % faster without the stream beyond it (1.5-2.4 M
    // nodes); the standalone flux launch gains at every size (bench level: 16.0-16.15 against 16.8-16.9 us).
    // With MGCFD_OPT_EXACT = 0 auto also takes the order-free kernel (bit 6) wherever it is the faster one: for every standalone
    // flux launch of a level that has the half-row plan (bench level 14.8 against 15.4 us, mixed-element level 16.4 against 17.5,
    // tetrahedral level 12.9 against 24.0), and for the fused stages of levels with long rows or halos beyond the shared table
    // (tetrahedral level: 45 against 78 us per sweep; on lattice-like levels the role-specialised node-gather stages stay ahead).
    int variant_for(const DeviceLevel &lv, bool fused = false) const
    {
        if (opt_variant >= 0) return opt_variant;
        // (round 4: the order-free stages are role-specialised where a level is on the kernel's fast path — halos within the
        //  smaller LDS image, at most five half rows per lane — and lead there too: bench level 47.1 against 51.1 us per sweep,
        //  the 4-level V-cycle 0.256 against 0.271 ms, profiles/r4_fast_mode.txt)
        const bool free_fast_path = !lv.dp.free_wide && lv.dp.halo_max <= kFreeCap4Halo && lv.dp.hr_max_rows <= kHalfMaxRows;
        if (!opt_exact && lv.dp.free_rows && (!fused || lv.dp.free_wide || lv.dp.has_tail || free_fast_path)) return 1 | 64;
        return 1;
    }
    // classes: bit0 internal, bit1 solid wall (-1), bit2 far field (-2)
    void op_flux(int l, int classes)
    {
        DeviceLevel &lv = level(l);
        Timed t(this, l, MGCFD_LOOP_FLUX, (classes & 1) != 0);
        if (classes != 7) settle_fluxes(lv);                // a partial launch leaves other nodes' memory as it is
        const int accumulate = lv.fluxes_zero ? 0 : 1;     // 0.0 + x: same bits either way
        const int variant = variant_for(lv);
        if ((variant & 4) && !lv.dp.edge_flux)              // two-phase design point: edge-flux scratch on first use
            lv.dp.edge_flux = dev_alloc<double>(static_cast<size_t>(lv.dp.n_edges_pad) * 5 + 8);
        if (opt_exact) exact::launch_flux(stream, lv.dp, lv.q, ff, lv.fluxes, classes, accumulate, variant, nullptr);
        else fast::launch_flux(stream, lv.dp, lv.q, ff, lv.fluxes, classes, accumulate, variant, nullptr);
        lv.fluxes_zero = false;
        lv.fluxes_stale = false;
        if (classes & 1) lv.iters[MGCFD_LOOP_FLUX] += lv.info.n_internal;
    }
    // One whole Runge-Kutta stage in one launch: fluxes of all edge classes from `in`, then
    // time_step into `out` (in != out).  fluxes[] stays logically zero, as after time_step.
    // apply_min: 0 no, 1 from the workgroups' partial minima, 2 from the (all-reduced) scalar
    // old: where the sweep's start state is read from (default: old_variables); look_ahead: the stage leaves
    // the next sweep's step-factor work behind (partial minima in partial_min, or fvcorr's factors in sf_alt)
    void op_fused_stage(int l, int j, const double *in, double *out, int apply_min, bool with_residual,
                        const double *old = nullptr, bool look_ahead = false, bool sumsq = false,
                        const double *vin_flux = nullptr, const int32_t *tile_list = nullptr, int32_t n_list = 0,
                        bool count_iters = true, const double *min_list = nullptr, int n_min = 0, bool lazy_residual = false,
                        const StagePush *push = nullptr)
    {
        DeviceLevel &lv = level(l);
        const FusedStep fs = fused_step(l, j, out, apply_min, with_residual, old, look_ahead, sumsq, vin_flux, tile_list, n_list, min_list, n_min, lazy_residual);
        Timed t(this, l, MGCFD_LOOP_FLUX, true);
        if (opt_exact) exact::launch_flux(stream, lv.dp, in, ff, lv.fluxes, 7, 0, variant_for(lv, true), &fs, push);
        else fast::launch_flux(stream, lv.dp, in, ff, lv.fluxes, 7, 0, variant_for(lv, true), &fs, push);
        if (count_iters) {                          // (a stage launched in two parts counts once)
            lv.iters[MGCFD_LOOP_FLUX] += lv.info.n_internal;
            lv.iters[MGCFD_LOOP_TIME_STEP] += lv.info.nel;
        }
    }
    // the arguments of one fused stage (what op_fused_stage launches; tools/exp/sweep_flow.patch takes three of them for one launch)
    FusedStep fused_step(int l, int j, double *out, int apply_min, bool with_residual, const double *old, bool look_ahead, bool sumsq,
                         const double *vin_flux, const int32_t *tile_list, int32_t n_list, const double *min_list, int n_min, bool lazy_residual)
    {
        DeviceLevel &lv = level(l);
        settle_residuals(lv);                       // (a sweep that supersedes the residual has dropped the flag: smooth_once)
        FusedStep fs;
        fs.tile_list = tile_list;
        fs.n_list = n_list;
        // the C++ partitioned sweeps (tile lists): ghosts numbered last are left to the halo messages
        if (tile_list && lv.plan.ghosts_last && lv.n_owned < lv.info.nel) fs.nel_active = lv.n_owned;
        fs.rk_div = double(MGCFD_RK + 1 - j);
        fs.step_factors = lv.step_factors;
        fs.old_variables = old ? old : lv.old_variables;
        fs.q_out = out;
        fs.next_partial_min = (look_ahead && mesh_variant != MGCFD_MESH_FVCORR) ? lv.partial_min : nullptr;
        fs.next_legacy_sf = (look_ahead && mesh_variant == MGCFD_MESH_FVCORR) ? lv.sf_alt : nullptr;
        fs.cbrt_vol = lv.cbrt_vol;
        if (out == lv.q) lv.min_ahead = false;      // (a caller that looks ahead sets it after its last stage)
        fs.partial_min = apply_min == 1 ? lv.partial_min : (apply_min == 2 ? lv.min_dt : (apply_min == 3 ? min_list : nullptr));
        fs.n_partial = apply_min == 2 ? 1 : (apply_min == 3 ? n_min : static_cast<int>((lv.info.nel + 255) / 256));
        fs.volumes = lv.volumes;
        fs.residuals = (with_residual && !lazy_residual) ? lv.residuals : nullptr;
        fs.sumsq_partial = (with_residual && sumsq) ? lv.tile_sumsq : nullptr;
        fs.vin_flux = vin_flux;                     // role 5: the input is the first stage's time_step of (old, vin_flux)
        fs.vin_div = double(MGCFD_RK + 1);
        fs.old_of_new = lv.dp.old_of_new;
        fs.err = err;
        if (vin_flux) fs.check_vin = next_check();       // the absorbed first stage's check_for_invalid_variables comes first
        fs.check = force_check >= 0 ? force_check : next_check();
        return fs;
    }
    void op_indirect_rw(int l)
    {
        DeviceLevel &lv = level(l);
        settle_fluxes(lv);
        Timed t(this, l, MGCFD_LOOP_INDIRECT_RW);
        if (opt_exact) exact::launch_indirect_rw(stream, lv.dp, lv.q, lv.fluxes, variant_for(lv));
        else fast::launch_indirect_rw(stream, lv.dp, lv.q, lv.fluxes, variant_for(lv));
        lv.fluxes_zero = false;
        lv.iters[MGCFD_LOOP_INDIRECT_RW] += lv.info.n_internal;
    }
    void op_zero_fluxes(int l)
    {
        DeviceLevel &lv = level(l);
        HIP_CHECK(hipMemsetAsync(lv.fluxes, 0, sizeof(double) * 5 * lv.dp.stride, stream));
        lv.fluxes_zero = true;
        lv.fluxes_stale = false;
    }
    // apply_min: 0 no, 1 from the workgroups' partial minima, 2 from the (all-reduced) scalar
    // old / out: default old_variables / variables (in place, as the reference); a sweep that keeps its
    // start state in `variables` passes both
    void op_time_step(int l, int j, int apply_min = 0, bool with_residual = false, bool lazy_zero = false,
                      const double *old = nullptr, double *out = nullptr)
    {
        if (j < 0 || j >= MGCFD_RK) throw std::invalid_argument("RK stage out of range");
        DeviceLevel &lv = level(l);
        settle_fluxes(lv);
        settle_residuals(lv);
        if (!old) old = lv.old_variables;
        if (!out) out = lv.q;
        if (out == lv.q) lv.min_ahead = false;
        Timed t(this, l, MGCFD_LOOP_TIME_STEP);
        const double *pm = apply_min == 1 ? lv.partial_min : (apply_min == 2 ? lv.min_dt : nullptr);
        const int n_pm = apply_min == 2 ? 1 : static_cast<int>((lv.info.nel + 255) / 256);
        double *res = with_residual ? lv.residuals : nullptr;
        const int check = next_check();
        if (opt_exact) exact::launch_time_step(stream, lv.info.nel, lv.dp.stride, j, lv.step_factors, lv.fluxes, old, out, lv.dp.old_of_new, err, check, pm, n_pm, lv.volumes, res, lazy_zero ? 0 : 1);
        else fast::launch_time_step(stream, lv.info.nel, lv.dp.stride, j, lv.step_factors, lv.fluxes, old, out, lv.dp.old_of_new, err, check, pm, n_pm, lv.volumes, res, lazy_zero ? 0 : 1);
        lv.fluxes_stale = lazy_zero;
        lv.fluxes_zero = true;
        lv.iters[MGCFD_LOOP_TIME_STEP] += lv.info.nel;
    }
    void op_residual(int l)
    {
        DeviceLevel &lv = level(l);
        lv.residuals_stale = false;
        if (opt_exact) exact::launch_residual(stream, lv.dp.stride, lv.old_variables, lv.q, lv.residuals);
        else fast::launch_residual(stream, lv.dp.stride, lv.old_variables, lv.q, lv.residuals);
    }
    void op_sumsq(int l)
    {
        DeviceLevel &lv = level(l);
        settle_residuals(lv);
        if (opt_exact) exact::launch_sumsq(stream, lv.info.nel, lv.dp.stride, lv.residuals, lv.partials, lv.n_partials, lv.sumsq, lv.dp.old_of_new, lv.n_owned);
        else fast::launch_sumsq(stream, lv.info.nel, lv.dp.stride, lv.residuals, lv.partials, lv.n_partials, lv.sumsq, lv.dp.old_of_new, lv.n_owned);
    }
    // rms (may be null): per-tile sums of squares of the fine level that the launch adds up on the side (cycle_once)
    void op_restrict(int fine, const SumTask *rms = nullptr)
    {
        DeviceLevel &F = level(fine);
        DeviceLevel &C = level(fine + 1);
        if (!F.has_transfer) throw std::invalid_argument("level has no multigrid map");
        settle_residuals(C);
        // Timer / iteration attribution quirk: the reference bumps `level` before the call,
        // so restriction is booked to the COARSE level.
        // the coarse sweep that follows starts with compute_step_factor on the restricted state: the
        // kernel leaves its first half (per-workgroup minima) in partial_min (global time step only)
        // (not on a partitioned level: its ghosts are stale until the halo exchange that follows)
        const bool ahead = mesh_variant != MGCFD_MESH_FVCORR && C.n_owned == C.info.nel;
        double *pm = ahead ? C.partial_min : nullptr;
        Timed t(this, fine + 1, MGCFD_LOOP_RESTRICT);
        const SumTask task = rms ? *rms : SumTask{};
        if (opt_exact) exact::launch_restrict(stream, C.info.nel, C.dp.stride, F.dp.stride, F.dp.child_ptr, F.dp.child, F.dp.child4, F.q, C.q, C.cbrt_vol, pm, task);
        else fast::launch_restrict(stream, C.info.nel, C.dp.stride, F.dp.stride, F.dp.child_ptr, F.dp.child, F.dp.child4, F.q, C.q, C.cbrt_vol, pm, task);
        C.min_ahead = ahead;
        C.iters[MGCFD_LOOP_RESTRICT] += 2 * F.info.mgc + C.info.nel;   // mg_loops.cpp:61,117,172
    }
    void op_prolong(int fine)
    {
        DeviceLevel &F = level(fine);
        DeviceLevel &C = level(fine + 1);
        if (!F.has_transfer) throw std::invalid_argument("level has no multigrid map");
        settle_residuals(C);
        settle_residuals(F);
        const bool ahead = mesh_variant != MGCFD_MESH_FVCORR && F.n_owned == F.info.nel;   // as in op_restrict
        double *pm = ahead ? F.partial_min : nullptr;
        Timed t(this, fine, MGCFD_LOOP_PROLONG);
        if (opt_exact) exact::launch_prolong(stream, F.dp, C.dp.stride, C.residuals, F.residuals, F.q, F.cbrt_vol, pm);
        else fast::launch_prolong(stream, F.dp, C.dp.stride, C.residuals, F.residuals, F.q, F.cbrt_vol, pm);
        F.min_ahead = ahead;
        F.iters[MGCFD_LOOP_PROLONG] += F.info.n_internal + F.info.nel;  // mg_loops.cpp:728,842
    }
    // Synchronises.  seq (may be null): the sequence number of the checked launch that found it.
    int read_error(int64_t *bad_cell, int *seq = nullptr)
    {
        unsigned long long h = 0;
        HIP_CHECK(hipMemcpyAsync(&h, err, sizeof(h), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        check_seq = 0;                              // (nothing is in flight: later launches count from 1 again)
        if (h == ~0ULL) return MGCFD_OK;
        invalid_cell = static_cast<int64_t>((h >> 8) & 0xFFFFFFFFULL);
        invalid_cycle = -1;
        if (bad_cell) *bad_cell = invalid_cell;
        if (seq) *seq = static_cast<int>(h >> 40);
        HIP_CHECK(hipMemsetAsync(err, 0xFF, sizeof(unsigned long long), stream));
        switch (h & 0xFF) {
            case 1: return MGCFD_ERR_NAN;
            case 2: return MGCFD_ERR_NEG_DENSITY;
            default: return MGCFD_ERR_NEG_ENERGY;
        }
    }
};

// Ranks in different processes, direct stores (mgcfd_rank_ipc_*): a wait for a neighbour's message that gave up (k_flags_wait,
// about two seconds) let the stages behind it run on stale ghosts.  Every call that synchronises looks at the counter and
// FAILS while it is not zero; mgcfd_rank_ipc_status reads it and thereby acknowledges.  Call with the stream idle.
static void fail_on_ipc_timeouts(mgcfd_solver *s)
{
    for (DeviceLevel &lv : s->L) {
        if (!lv.hx || !lv.hx->ipc || !lv.hx->ipc_timeouts) continue;
        int n = 0;
        HIP_CHECK(hipMemcpy(&n, lv.hx->ipc_timeouts, sizeof(int), hipMemcpyDeviceToHost));
        if (n != 0) {
            lv.hx->ipc_unacknowledged = true;
            throw HipError(std::to_string(n) + " wait(s) for a neighbouring rank's message gave up: the state of this rank was computed from stale ghosts "
                           "(mgcfd_rank_ipc_status acknowledges; mgcfd_rank_ipc_detach returns to the buffered exchange)");
        }
    }
}

mgcfd_solver::~mgcfd_solver()
{
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    for (auto &g : sweep_graphs) if (g.second.exec) (void)hipGraphExecDestroy(g.second.exec);
    for (auto &g : cycle_graphs) if (g.second.exec) (void)hipGraphExecDestroy(g.second.exec);
    if (rms_ring) (void)hipFree(rms_ring);
    if (rms_count) (void)hipFree(rms_count);
    for (auto &p : pending) { (void)hipEventDestroy(p.start); (void)hipEventDestroy(p.stop); }
    for (auto e : free_events) (void)hipEventDestroy(e);
    for (auto &lv : L) {
        void *ptrs[] = {lv.q_alt, lv.sf_alt, lv.tile_sumsq, lv.dp.nbr16, lv.dp.tile_halo, lv.dp.tile_ovf_ptr, lv.dp.tile_ovf, lv.q, lv.old_variables, lv.fluxes, lv.residuals, lv.step_factors, lv.volumes,
                        lv.cbrt_vol, lv.min_dt, lv.partial_min, lv.sumsq, lv.partials, lv.dp.slice_row0, lv.dp.rows_int,
                        lv.dp.rows_bnd, lv.dp.nbr, lv.dp.w, lv.dp.old_of_new, lv.dp.child_ptr, lv.dp.child, lv.dp.child4, lv.dp.pro_w, lv.dp.pro_p, lv.dp.pro_tile_n, lv.dp.pro_tile_ids, lv.dp.pro_s16, lv.dp.pro_own16,
                        lv.dp.pro_parent, lv.dp.pro_wsum, lv.dp.te_chunk_ptr, lv.dp.te_count, lv.dp.te_slots, lv.dp.te_w, lv.dp.te_w3, lv.dp.hr_row0, lv.dp.hr_code, lv.dp.hr_w, lv.dp.hg16, lv.dp.free_halo,
                        lv.dp.gat16, lv.dp.fe_ab, lv.dp.fe_w, lv.dp.row_edge, lv.dp.edge_flux,
                        const_cast<int32_t *>(lv.dp.tail.rows_main), const_cast<int32_t *>(lv.dp.tail.tile_ptr),
                        const_cast<double2 *>(lv.dp.tail.rec), const_cast<int32_t *>(lv.dp.tail.begin),
                        const_cast<int32_t *>(lv.dp.tail.count), lv.dp.tail.flux};
        for (void *p : ptrs) if (p && !lv.in_block(p)) (void)hipFree(p);      // (what an option uploaded later has an allocation of its own)
        if (lv.block) (void)hipFree(lv.block);
        for (auto &hp : lv.halo_plans) if (hp.first) (void)hipFree(hp.first);
        if (lv.hx) {
            HaloExchange &hx = *lv.hx;
            for (void *m : hx.ipc_opened) (void)hipIpcCloseMemHandle(m);
            void *hp[] = {hx.node_send_ptr, hx.node_send_target, hx.node_send_peer, hx.send_idx, hx.recv_idx, hx.tiles_boundary, hx.tiles_interior, hx.tiles_all, hx.gmin, hx.peer_scalars[0], hx.peer_scalars[1], hx.min_par, hx.push_target, hx.flags, hx.ticket, hx.ipc_timeouts, hx.gmins};
            for (hipEvent_t e : hx.bdone) if (e) (void)hipEventDestroy(e);
            for (void *p : hp) if (p) (void)hipFree(p);
            for (int b = 0; b < HaloExchange::kSets; b++) {
                if (hx.send_buf[b]) (void)hipFree(hx.send_buf[b]);
                if (hx.recv_buf[b]) (void)hipFree(hx.recv_buf[b]);
                if (hx.packed[b]) (void)hipEventDestroy(hx.packed[b]);
                if (hx.arrived[b]) (void)hipEventDestroy(hx.arrived[b]);
            }
            for (hipGraphExec_t ge : hx.sweep_graph) if (ge) (void)hipGraphExecDestroy(ge);
            for (hipEvent_t e : {hx.reduced, hx.gathered, hx.joined}) if (e) (void)hipEventDestroy(e);
            if (hx.comm_stream) (void)hipStreamDestroy(hx.comm_stream);
        }
    }
    if (err) (void)hipFree(err);
    if (own_stream) (void)hipStreamDestroy(own_stream);
}

// ------------------------------------------------------------------------------------------
// construction
// ------------------------------------------------------------------------------------------
// mgcfd_device_warm_up: the runtime's and the device context's start-up on a thread of its own (once per process)
static std::mutex g_warm_mutex;
static std::shared_future<void> g_warm;
static bool warm_up_under_way()
{
    std::shared_future<void> f;
    { std::lock_guard<std::mutex> lk(g_warm_mutex); f = g_warm; }
    return f.valid() && f.wait_for(std::chrono::seconds(0)) != std::future_status::ready;
}
static void wait_for_warm_up()
{
    std::shared_future<void> f;
    { std::lock_guard<std::mutex> lk(g_warm_mutex); f = g_warm; }
    if (f.valid()) f.wait();
}

static std::unique_ptr<mgcfd_solver> build_solver(const mgcfd_level_desc *levels, int nlevels, int mesh_variant, int device,
                                                  const int64_t *n_owned = nullptr, const int64_t *const *order_keys = nullptr)
{
    if (!levels || nlevels <= 0) throw std::invalid_argument("no levels given");
    if (mesh_variant != MGCFD_MESH_FVCORR && mesh_variant != MGCFD_MESH_M6_WING &&
        mesh_variant != MGCFD_MESH_LA_CASCADE && mesh_variant != MGCFD_MESH_ROTOR_37)
        throw std::invalid_argument("unknown mesh variant");
    // The device is looked at before any host work, unless a warm-up of it is still under way on its own thread
    // (mgcfd_device_warm_up): then the gather plans, which need no device, are built beside it and the device's turn
    // comes before the uploads.
    auto device_ready = [&]() {
        wait_for_warm_up();
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            throw HipError("no HIP device available: libmgcfd_hip.so has no CPU fallback");
        if (device < 0 || device >= ndev) throw std::invalid_argument("device index out of range");
    };
    const bool device_later = warm_up_under_way();
    if (!device_later) device_ready();

    auto s = std::make_unique<mgcfd_solver>();
    s->device = device;
    s->mesh_variant = mesh_variant;
    far_field_constants(s->ff17);
    std::memcpy(s->ff.var, s->ff17, sizeof(double) * 5);
    std::memcpy(s->ff.fc_mx, s->ff17 + 5, sizeof(double) * 3);
    std::memcpy(s->ff.fc_my, s->ff17 + 8, sizeof(double) * 3);
    std::memcpy(s->ff.fc_mz, s->ff17 + 11, sizeof(double) * 3);
    std::memcpy(s->ff.fc_de, s->ff17 + 14, sizeof(double) * 3);

    s->L.resize(static_cast<size_t>(nlevels));
    const bool timing = std::getenv("MGCFD_PLAN_TIMING") != nullptr;      // where the host time of a solver's creation goes (stderr)
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[mgcfd create] %-34s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    PlanOptions popt;
    if (const char *o = std::getenv("MGCFD_ORDERING")) popt.ordering = std::atoi(o);   // diagnostic override
    if (const char *o = std::getenv("MGCFD_TILE_ORDER")) popt.tile_order = std::atoi(o);   // (A/B: 0 as clustered, 1 costliest first, 2 default)
    if (const char *o = std::getenv("MGCFD_TILE_CURVE")) popt.tile_curve = std::atoi(o);   // (A/B: 0 as clustered, 1 Morton — default —, 2 Hilbert)
    // host-side plans first (coarse permutations are needed by the fine level's transfer plan).  The levels' plans are
    // independent of each other: one host thread per level (level 0 of the M6-like hierarchy takes 1 s, all four 2.2 s in a row).
    auto run_per_level = [&](int count, auto &&body) {
        std::vector<std::exception_ptr> errors(static_cast<size_t>(std::max(count, 0)));
        std::vector<std::thread> workers;
        auto guarded_body = [&](int l) { try { body(l); } catch (...) { errors[static_cast<size_t>(l)] = std::current_exception(); } };
        for (int l = 1; l < count; l++) workers.emplace_back(guarded_body, l);
        if (count > 0) guarded_body(0);
        for (std::thread &w : workers) w.join();
        for (const std::exception_ptr &e : errors) if (e) std::rethrow_exception(e);      // (the lowest level's error first)
    };
    run_per_level(nlevels, [&](int l) {
        PlanOptions popt_l = popt;                           // (a thread's own copy: n_owned differs per level)
        PlanOptions &popt = popt_l;
        const mgcfd_level_desc &d = levels[l];
        DeviceLevel &lv = s->L[static_cast<size_t>(l)];
        if (!d.volumes || !d.edges) throw std::invalid_argument("level is missing volumes/edges");
        if (d.internal_start < 0 || d.internal_start + d.n_internal > d.n_edges || d.boundary_start < 0 ||
            d.boundary_start + d.n_boundary > d.n_edges || d.wall_start < 0 || d.wall_start + d.n_wall > d.n_edges)
            throw std::invalid_argument("edge class ranges exceed n_edges");
        lv.edges.assign(d.edges, d.edges + d.n_edges);
        adjust_and_dampen(d, mesh_variant, lv.edges);
        lv.n_owned = d.nel;
        popt.n_owned = -1;
        if (n_owned && n_owned[l] >= 0 && n_owned[l] < d.nel) {
            lv.n_owned = n_owned[l];
            popt.n_owned = n_owned[l];
        }
        popt.long_rows = !std::getenv("MGCFD_NO_LONG_ROWS");      // (diagnostic: per-node rows only, overflow entries gathered in the loop)
        build_level_plan(d, lv.edges, popt, lv.plan);
        if (std::getenv("MGCFD_VERBOSE"))
            std::fprintf(stderr, "[mgcfd] level %d: %ld nodes, %d tiles, halo mean %.0f max %d (cap %d), overflow refs %ld, ELL padding %.1f%%, tile edges mean %.0f max %d (edge-once %s)\n",
                         l, (long)d.nel, lv.plan.n_tiles, lv.plan.halo_mean, lv.plan.halo_max, kTileCap - kTile,
                         (long)lv.plan.halo_overflow_refs, 100.0 * lv.plan.pad_fraction, lv.plan.te_mean, lv.plan.te_max,
                         lv.plan.edge_once ? "yes" : "no");
        if (std::getenv("MGCFD_VERBOSE"))
            std::fprintf(stderr, "[mgcfd] level %d: half rows %s: %ld evaluations (%.2f per node), %ld in another node's lane, %ld padding slots (%.1f%%)\n", l, lv.plan.half ? "yes" : "no",
                         (long)lv.plan.hr_entries, double(lv.plan.hr_entries) / double(d.nel), (long)lv.plan.hr_foreign, (long)lv.plan.hr_padding,
                         lv.plan.hr_entries ? 100.0 * double(lv.plan.hr_padding) / double(lv.plan.hr_entries) : 0.0);
        if (std::getenv("MGCFD_VERBOSE")) {
            // how many staged halo nodes belong to a tile of ANOTHER XCD's range (their state cannot meet its owner's in an L2)
            const int64_t nt = lv.plan.n_tiles, q8 = nt >> 3, r8 = nt & 7;
            auto xcd_of = [&](int64_t t) { return t < r8 * (q8 + 1) ? t / (q8 + 1) : r8 + (t - r8 * (q8 + 1)) / std::max<int64_t>(q8, 1); };
            int64_t total = 0, foreign = 0, near_ = 0;
            for (int64_t t = 0; t < nt; t++)
                for (int32_t k = lv.plan.tile_halo_ptr[static_cast<size_t>(t)]; k < lv.plan.tile_halo_ptr[static_cast<size_t>(t) + 1]; k++) {
                    const int64_t o = lv.plan.tile_halo[static_cast<size_t>(k)] / kTile;
                    total++;
                    if (xcd_of(o) != xcd_of(t)) foreign++;
                    else if (std::llabs(o - t) <= 48) near_++;
                }
            std::fprintf(stderr, "[mgcfd] level %d: staged halo nodes %ld: %.1f%% owned by a tile of another XCD's range, %.1f%% by a tile within 48 of the reader in its own range\n",
                         l, (long)total, total ? 100.0 * foreign / total : 0.0, total ? 100.0 * near_ / total : 0.0);
        }
        if (std::getenv("MGCFD_VERBOSE") && lv.plan.has_tail) {
            int64_t rows_full = 0, rows_cut = 0;
            for (size_t q = 0; q < lv.plan.rows_int.size(); q++) { rows_full += lv.plan.rows_int[q]; rows_cut += lv.plan.rows_main[q]; }
            std::fprintf(stderr, "[mgcfd] level %d: long rows: %ld entries left to the workgroups (%.1f per tile), per-node loop %.1f -> %.1f rows per slice\n",
                         l, (long)lv.plan.tail_total, double(lv.plan.tail_total) / lv.plan.n_tiles,
                         double(rows_full) / double(lv.plan.rows_int.size()), double(rows_cut) / double(lv.plan.rows_int.size()));
        }
    });
    lap("level plans (a thread per level)");
    // (a transfer plan writes its own fine level's plan and only READS the coarse level's permutation)
    run_per_level(nlevels - 1, [&](int l) {
        const mgcfd_level_desc &d = levels[l];
        if (!d.mg_map) throw std::invalid_argument("multigrid map missing between levels");
        build_transfer_plan(d, s->L[static_cast<size_t>(l)].edges, levels[l + 1].coords, levels[l + 1].nel,
                            s->L[static_cast<size_t>(l) + 1].plan.new_of_old, s->L[static_cast<size_t>(l)].plan,
                            order_keys ? order_keys[l] : nullptr, s->L[static_cast<size_t>(l)].n_owned);
        s->L[static_cast<size_t>(l)].has_transfer = true;
    });
    lap("transfer plans");
    // What the device gets, level by level, listed and repacked with no call into the runtime (LevelStaging): a host thread
    // per level.  What only a non-default option reaches — the order-free / half-row plan, the edge-once lists, the
    // two-phase arrays, the 32-bit neighbour codes — stays on the host until an option asks for it (upload_optional_plans):
    // 60 % of the bytes.
    std::vector<LevelStaging> staging(static_cast<size_t>(nlevels));
    run_per_level(nlevels, [&](int l) {
        LevelStaging &st = staging[static_cast<size_t>(l)];
        const mgcfd_level_desc &d = levels[l];
        DeviceLevel &lv = s->L[static_cast<size_t>(l)];
        lv.info = d;
        lv.info.volumes = nullptr; lv.info.coords = nullptr; lv.info.edges = nullptr; lv.info.mg_map = nullptr;
        const int64_t nel = d.nel;
        const LevelPlan &P = lv.plan;
        const int64_t stride = int64_t(P.n_slices) * kSlice;
        std::vector<double> vol(static_cast<size_t>(stride), 1.0), cb(static_cast<size_t>(stride), 1.0);
        for (int64_t n = 0; n < nel; n++) {
            const double v = d.volumes[P.old_of_new[static_cast<size_t>(n)]];
            vol[static_cast<size_t>(n)] = v;
            cb[static_cast<size_t>(n)] = std::cbrt(v);        // cfd_loops.cpp:116, static => host libm once
            // a ghost never holds the minimum of compute_step_factor (its owner counts it): a launch that looks ahead
            // sees a ghost at the sweep's START state, which must not enter the next sweep's minimum
            if (P.old_of_new[static_cast<size_t>(n)] >= lv.n_owned) cb[static_cast<size_t>(n)] = std::numeric_limits<double>::infinity();
        }
        st.upload(lv.volumes, std::move(vol));
        st.upload(lv.cbrt_vol, std::move(cb));
        st.alloc(lv.fluxes, sizeof(double) * (static_cast<size_t>(stride) * 5));
        st.alloc(lv.residuals, sizeof(double) * (static_cast<size_t>(stride) * 5));
        st.alloc(lv.step_factors, sizeof(double) * (static_cast<size_t>(stride)));
        st.alloc(lv.sf_alt, sizeof(double) * (static_cast<size_t>(stride)));
        st.alloc(lv.min_dt, sizeof(double) * (1));
        // (+inf: a tile that is never launched — the ghost-only tiles of a partitioned level — holds no minimum)
        st.upload(lv.partial_min, std::vector<double>(static_cast<size_t>((nel + 255) / 256), std::numeric_limits<double>::infinity()));
        st.alloc(lv.tile_sumsq, sizeof(double) * (static_cast<size_t>((nel + 255) / 256)));
        st.alloc(lv.sumsq, sizeof(double) * (1));
        lv.n_partials = static_cast<int>(std::min<int64_t>(1024, (nel * 5 + 255) / 256));
        st.alloc(lv.partials, sizeof(double) * (static_cast<size_t>(lv.n_partials)));
        lv.dp.nel = nel;
        lv.dp.stride = stride;
        lv.dp.n_slices = P.n_slices;
        st.upload(lv.dp.slice_row0, P.slice_row0);
        st.upload(lv.dp.rows_int, P.rows_int);
        st.upload(lv.dp.rows_bnd, P.rows_bnd);
        {
            // edge weights as [row][component][lane] so each component load of a wave is one
            // contiguous 512-byte run
            std::vector<double> ws(P.w.size() * 4 + 2 * 4 * kSlice, 0.0);   // + two rows of padding (k_flux_tile's prologue)
            for (size_t e = 0; e < P.w.size(); e++) {
                const size_t row = e / kSlice, lane = e % kSlice;
                ws[(row * 4 + 0) * kSlice + lane] = P.w[e].x;
                ws[(row * 4 + 1) * kSlice + lane] = P.w[e].y;
                ws[(row * 4 + 2) * kSlice + lane] = P.w[e].z;
                ws[(row * 4 + 3) * kSlice + lane] = P.w[e].k;
            }
            st.upload(lv.dp.w, std::move(ws));
        }
        st.upload(lv.dp.old_of_new, P.old_of_new);
        lv.dp.n_tiles = P.n_tiles;
        lv.plan.nbr16.resize(P.nbr16.size() + 2 * kSlice, static_cast<uint16_t>(kT16Pad));   // two rows of padding
        st.upload(lv.dp.nbr16, P.nbr16);
        {
            std::vector<int32_t> fixed(static_cast<size_t>(P.n_tiles) * kHaloStride, -1);
            for (int32_t t = 0; t < P.n_tiles; t++)
                std::copy(P.tile_halo.begin() + P.tile_halo_ptr[static_cast<size_t>(t)],
                          P.tile_halo.begin() + P.tile_halo_ptr[static_cast<size_t>(t) + 1],
                          fixed.begin() + static_cast<size_t>(t) * kHaloStride);
            st.upload(lv.dp.tile_halo, std::move(fixed));
        }
        st.upload(lv.dp.tile_ovf_ptr, P.tile_ovf_ptr);
        st.upload(lv.dp.tile_ovf, P.tile_ovf);
        lv.row_bytes = int64_t(P.slice_row0.back()) * kSlice * 34;
        lv.dp.pad_row = P.slice_row0.back();
        lv.dp.pad_chunk = P.te_chunk_ptr.empty() ? 0 : P.te_chunk_ptr.back();
        lv.dp.n_edges = d.n_internal;
        lv.dp.n_edges_pad = (d.n_internal + 255) / 256 * 256;
        lv.plan.row_edge.resize(P.row_edge.size() + 2 * kSlice, -1);                     // two rows of padding
        lv.dp.has_tail = P.has_tail ? 1 : 0;
        if (lv.dp.has_tail) {
            st.upload(lv.dp.tail.rows_main, P.rows_main);
            st.upload(lv.dp.tail.tile_ptr, P.tail_tile_ptr);
            st.upload(lv.dp.tail.rec, P.tail_rec);
            lv.plan.tail_begin.resize(static_cast<size_t>(lv.dp.stride), 0);         // (threads past nel read these too)
            lv.plan.tail_count.resize(static_cast<size_t>(lv.dp.stride), 0);
            st.upload(lv.dp.tail.begin, P.tail_begin);
            st.upload(lv.dp.tail.count, P.tail_count);
            st.alloc(lv.dp.tail.flux, sizeof(double) * static_cast<size_t>(6 * P.tail_total));
        }
        lv.dp.vin_ok = (P.halo_overflow_refs == 0 && P.halo_max <= kTile) ? 1 : 0;
        lv.dp.lds_complete = P.halo_overflow_refs == 0 ? 1 : 0;
        lv.dp.halo_max = P.halo_max;
        lv.dp.edge_once = (P.edge_once && !std::getenv("MGCFD_NO_EDGE_ONCE")) ? 1 : 0;
        if (lv.dp.edge_once) {
            lv.plan.te_slots.resize(P.te_slots.size() + 2 * kEdgeChunk, static_cast<uint16_t>(kT16Pad));   // one chunk of padding
            lv.plan.te_w.resize(P.te_w.size() + 4 * kEdgeChunk, 0.0);
            lv.plan.gat16.resize(P.gat16.size() + 2 * kSlice, static_cast<uint16_t>(kT16Pad));   // two rows of padding
        }
        lv.dp.half = (P.half && !std::getenv("MGCFD_NO_HALF_ROWS")) ? 1 : 0;
        lv.dp.free_rows = (P.free_rows && !std::getenv("MGCFD_NO_HALF_ROWS")) ? 1 : 0;
        lv.dp.hr_max_rows = P.hr_max_rows;
        lv.dp.free_wide = (lv.dp.free_rows && P.free_wide) ? 1 : 0;
        if (!lv.dp.free_wide) { lv.plan.free_halo.clear(); lv.plan.free_halo.shrink_to_fit(); }
        if (lv.dp.free_rows) {
            lv.dp.hr_pad_row = P.hr_row0.back();
            lv.plan.hr_code.resize(P.hr_code.size() + 2 * kSlice, kHalfPad);      // two half rows of padding
            lv.plan.hr_w.resize(P.hr_w.size() + 2 * 3 * kSlice, 0.0);
            lv.plan.hg16.resize(P.hg16.size() + 2 * kSlice, static_cast<uint16_t>(kT16Pad));
        } else {
            lv.plan.hr_code.clear(); lv.plan.hr_code.shrink_to_fit();
            lv.plan.hr_w.clear(); lv.plan.hr_w.shrink_to_fit();
            lv.plan.hg16.clear(); lv.plan.hg16.shrink_to_fit();
        }
        if (!lv.dp.edge_once) {
            lv.plan.te_slots.clear(); lv.plan.te_slots.shrink_to_fit();
            lv.plan.te_w.clear(); lv.plan.te_w.shrink_to_fit();
            lv.plan.gat16.clear(); lv.plan.gat16.shrink_to_fit();
        }
        if (lv.has_transfer) {
            st.upload(lv.dp.child_ptr, P.child_ptr);
            st.upload(lv.dp.child, P.child);
            {
                const size_t nc = P.child_ptr.size() - 1;
                std::vector<int32_t> c4(nc * 4, -1);
                for (size_t c = 0; c < nc; c++)
                    for (int32_t k = P.child_ptr[c]; k < P.child_ptr[c + 1] && k < P.child_ptr[c] + 4; k++)
                        c4[c * 4 + static_cast<size_t>(k - P.child_ptr[c])] = P.child[static_cast<size_t>(k)];
                st.upload(lv.dp.child4, std::move(c4));
            }
            {
                // prolongation entries as [row][component][lane]: every load of a wave is one contiguous run
                const size_t rows_n = P.pro.size() / kSlice;
                std::vector<double> pw((rows_n + 4) * 2 * kSlice, 0.0);      // + four rows of padding (k_prolong_tile reads ahead)
                std::vector<int32_t> pp(rows_n * kSlice);
                for (size_t e = 0; e < P.pro.size(); e++) {
                    const size_t row = e / kSlice, lane = e % kSlice;
                    pw[(row * 2 + 0) * kSlice + lane] = P.pro[e].w_own;
                    pw[(row * 2 + 1) * kSlice + lane] = P.pro[e].w_other;
                    pp[row * kSlice + lane] = P.pro[e].p_other;
                }
                st.upload(lv.dp.pro_w, std::move(pw));
                st.upload(lv.dp.pro_p, std::move(pp));
            }
            lv.dp.pro_tiled = P.pro_tiled ? 1 : 0;
            if (P.pro_tiled) {
                st.upload(lv.dp.pro_tile_n, P.pro_tile_n);
                st.upload(lv.dp.pro_tile_ids, P.pro_tile_ids);
                lv.plan.pro_s16.resize(P.pro_s16.size() + 4 * kSlice, 0);          // four rows of padding
                st.upload(lv.dp.pro_s16, P.pro_s16);
                st.upload(lv.dp.pro_own16, P.pro_own16);
            }
            st.upload(lv.dp.pro_parent, P.pro_parent);
            st.upload(lv.dp.pro_wsum, P.pro_wsum);
            // the per-entry host copies are not needed again
            lv.plan.pro.clear(); lv.plan.pro.shrink_to_fit();
        }
        lv.plan.w.clear(); lv.plan.w.shrink_to_fit();
    });
    lap("device layouts repacked (a thread per level)");
    if (device_later) device_ready();
    s->use_device();
    HIP_CHECK(hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking));
    s->stream = s->own_stream;
    s->err = dev_alloc<unsigned long long>(1);
    HIP_CHECK(hipMemset(s->err, 0xFF, sizeof(unsigned long long)));
    lap("device (behind its warm-up), stream");
    // the copies out of pageable memory are host work too: a thread per level again, every thread selects the device
    // for itself; the state's initialisation goes to the one stream
    run_per_level(nlevels, [&](int l) {
        s->use_device();
        DeviceLevel &lv = s->L[static_cast<size_t>(l)];
        LevelStaging &st = staging[static_cast<size_t>(l)];
        const int64_t stride = lv.dp.stride;
        HIP_CHECK(hipMalloc(&lv.block, st.total));
        lv.block_bytes = st.total;
        for (const LevelStaging::Item &it : st.items) {
            it.place(static_cast<char *>(lv.block));
            if (it.bytes) HIP_CHECK(hipMemcpy(static_cast<char *>(lv.block) + it.offset, it.src, it.bytes, hipMemcpyHostToDevice));
        }
        st = LevelStaging();
        // (the three state buffers keep allocations of their own: a partitioned level exports them to its peers'
        // processes, and an inter-process handle names a whole allocation)
        lv.q = dev_alloc<double>(static_cast<size_t>(stride) * kNumStateFields);
        lv.q_alt = dev_alloc<double>(static_cast<size_t>(stride) * 5);
        lv.old_variables = dev_alloc<double>(static_cast<size_t>(stride) * 5);
        lv.state[0] = lv.q; lv.state[1] = lv.q_alt; lv.state[2] = lv.old_variables;
        lv.sfb[0] = lv.step_factors; lv.sfb[1] = lv.sf_alt;
        lv.plan.nbr16.clear(); lv.plan.nbr16.shrink_to_fit();
        s->upload_optional_plans(lv);
        // initial state: far field everywhere, fluxes/residuals/old/step factors zero
        HIP_CHECK(hipMemsetAsync(lv.old_variables, 0, sizeof(double) * 5 * stride, s->stream));
        HIP_CHECK(hipMemsetAsync(lv.fluxes, 0, sizeof(double) * 5 * stride, s->stream));
        HIP_CHECK(hipMemsetAsync(lv.residuals, 0, sizeof(double) * 5 * stride, s->stream));
        HIP_CHECK(hipMemsetAsync(lv.step_factors, 0, sizeof(double) * stride, s->stream));
        HIP_CHECK(hipMemsetAsync(lv.sf_alt, 0, sizeof(double) * stride, s->stream));
        exact::launch_init_variables(s->stream, stride, s->ff, lv.q);
        exact::launch_init_variables(s->stream, stride, s->ff, lv.q_alt);     // valid numbers in the padded tail
    });
    s->use_device();
    HIP_CHECK(hipStreamSynchronize(s->stream));
    HIP_CHECK(hipGetLastError());
    lap("one block per level, copies (a thread per level)");
    return s;
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
template <typename F> static int guarded(F &&f)
{
    try {
        f();
        return MGCFD_OK;
    } catch (const HipError &e) {
        g_last_error = e.what();
        return MGCFD_ERR_HIP;
    } catch (const std::invalid_argument &e) {
        g_last_error = e.what();
        return MGCFD_ERR_ARG;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return MGCFD_ERR_IO;
    }
}

#define REQUIRE(p) do { if (!(p)) { g_last_error = "null argument: " #p; return MGCFD_ERR_ARG; } } while (0)

extern "C" {

const char *mgcfd_last_error(void) { return g_last_error.c_str(); }
int mgcfd_abi_version(void) { return 1; }
int mgcfd_device_warm_up(int device)
{
    std::lock_guard<std::mutex> lk(g_warm_mutex);
    if (!g_warm.valid())
        g_warm = std::async(std::launch::async, [device] { if (hipSetDevice(device) == hipSuccess) (void)hipFree(nullptr); }).share();
    return MGCFD_OK;
}

// ---- file boundary ----
int mgcfd_mesh_load(const char *input_dat, const char *directory, int duplicate, mgcfd_mesh **out)
{
    REQUIRE(input_dat); REQUIRE(out);
    return guarded([&] {
        auto m = std::make_unique<mgcfd_mesh>();
        m->mesh = load_mesh(input_dat, directory ? directory : "", duplicate);
        *out = m.release();
    });
}
int mgcfd_mesh_load_ex(const char *input_dat, const char *directory, int duplicate, int flags, mgcfd_mesh **out)
{
    REQUIRE(input_dat); REQUIRE(out);
    return guarded([&] {
        auto m = std::make_unique<mgcfd_mesh>();
        m->mesh = load_mesh(input_dat, directory ? directory : "", duplicate, (flags & MGCFD_MESH_LEGACY_ORDERING) != 0);
        *out = m.release();
    });
}
void mgcfd_mesh_free(mgcfd_mesh *m) { delete m; }
int mgcfd_mesh_num_levels(const mgcfd_mesh *m) { return m ? static_cast<int>(m->mesh.levels.size()) : 0; }
int mgcfd_mesh_variant(const mgcfd_mesh *m) { return m ? m->mesh.mesh_variant : -1; }
int mgcfd_mesh_size(const mgcfd_mesh *m) { return m ? m->mesh.size : 0; }
int mgcfd_mesh_level(const mgcfd_mesh *m, int level, mgcfd_level_desc *out)
{
    REQUIRE(m); REQUIRE(out);
    if (level < 0 || level >= static_cast<int>(m->mesh.levels.size())) { g_last_error = "level out of range"; return MGCFD_ERR_ARG; }
    *out = m->mesh.levels[static_cast<size_t>(level)].desc();
    return MGCFD_OK;
}
int mgcfd_write_array(const char *path, const double *data, int64_t nel, int ncols)
{
    REQUIRE(path); REQUIRE(data);
    return guarded([&] { write_array(path, data, nel, ncols); });
}
int mgcfd_identify_differences(const double *t, const double *m, int64_t nel, int mesh_variant, int64_t *first_bad)
{
    REQUIRE(t); REQUIRE(m);
    const int64_t k = identify_differences(t, m, nel, mesh_variant);
    if (first_bad) *first_bad = k;
    if (k >= 0) { g_last_error = "Unacceptable error detected at flat index " + std::to_string(k); return MGCFD_ERR_VALIDATION; }
    return MGCFD_OK;
}

// Host only: the gather plans mgcfd_create[_partitioned[_mg]] would build for these levels, audited (preprocess.cpp:
// audit_level_plan).  No device is touched.
int mgcfd_plan_audit(const mgcfd_level_desc *levels, int nlevels, int mesh_variant, const int64_t *n_owned,
                     const int64_t *const *order_keys, char *report, int64_t report_cap)
{
    REQUIRE(levels);
    std::string rep;
    const int rc = guarded([&] {
        if (nlevels <= 0) throw std::invalid_argument("no levels given");
        std::vector<LevelPlan> plans(static_cast<size_t>(nlevels));
        std::vector<std::vector<mgcfd_edge>> edges(static_cast<size_t>(nlevels));
        for (int l = 0; l < nlevels; l++) {
            const mgcfd_level_desc &d = levels[l];
            if (!d.volumes || !d.edges) throw std::invalid_argument("level is missing volumes/edges");
            edges[static_cast<size_t>(l)].assign(d.edges, d.edges + d.n_edges);
            adjust_and_dampen(d, mesh_variant, edges[static_cast<size_t>(l)]);
            PlanOptions popt;
            if (const char *o = std::getenv("MGCFD_TILE_ORDER")) popt.tile_order = std::atoi(o);
            if (const char *o = std::getenv("MGCFD_TILE_CURVE")) popt.tile_curve = std::atoi(o);
            if (n_owned && n_owned[l] >= 0 && n_owned[l] < d.nel) popt.n_owned = n_owned[l];
            build_level_plan(d, edges[static_cast<size_t>(l)], popt, plans[static_cast<size_t>(l)]);
        }
        for (int l = 0; l + 1 < nlevels; l++) {
            if (!levels[l].mg_map) throw std::invalid_argument("multigrid map missing between levels");
            build_transfer_plan(levels[l], edges[static_cast<size_t>(l)], levels[l + 1].coords, levels[l + 1].nel,
                                plans[static_cast<size_t>(l) + 1].new_of_old, plans[static_cast<size_t>(l)],
                                order_keys ? order_keys[l] : nullptr, (n_owned && n_owned[l] >= 0) ? n_owned[l] : -1);
        }
        for (int l = 0; l < nlevels; l++) {
            const std::string r = audit_level_plan(levels[l], plans[static_cast<size_t>(l)], l + 1 < nlevels ? levels[l + 1].nel : -1);
            if (!r.empty()) rep += "level " + std::to_string(l) + ":\n" + r;
            if (std::getenv("MGCFD_PLAN_DIGEST")) {          // (diagnostic: the report then always holds text)
                char buf[64];
                std::snprintf(buf, sizeof(buf), "digest level %d: %016llx\n", l, (unsigned long long)plan_digest(plans[static_cast<size_t>(l)]));
                rep += buf;
            }
        }
    });
    if (rc != MGCFD_OK) return rc;
    if (report && report_cap > 0) { std::snprintf(report, static_cast<size_t>(report_cap), "%s", rep.c_str()); }
    if (!rep.empty()) { g_last_error = "plan audit: " + rep; return MGCFD_ERR_ARG; }
    return MGCFD_OK;
}

// ---- life cycle ----
int mgcfd_create(const mgcfd_level_desc *levels, int nlevels, int mesh_variant, int device, mgcfd_solver **out)
{
    REQUIRE(out);
    return guarded([&] { *out = build_solver(levels, nlevels, mesh_variant, device).release(); });
}
int mgcfd_create_partitioned(const mgcfd_level_desc *levels, int nlevels, int mesh_variant, int device,
                             const int64_t *n_owned, mgcfd_solver **out)
{
    REQUIRE(out); REQUIRE(n_owned);
    return guarded([&] { *out = build_solver(levels, nlevels, mesh_variant, device, n_owned).release(); });
}
int mgcfd_create_partitioned_mg(const mgcfd_level_desc *levels, int nlevels, int mesh_variant, int device,
                                const int64_t *n_owned, const int64_t *const *order_keys, mgcfd_solver **out)
{
    REQUIRE(out); REQUIRE(n_owned);
    return guarded([&] { *out = build_solver(levels, nlevels, mesh_variant, device, n_owned, order_keys).release(); });
}
int mgcfd_create_from_mesh(const mgcfd_mesh *m, int device, mgcfd_solver **out)
{
    REQUIRE(m); REQUIRE(out);
    return guarded([&] {
        std::vector<mgcfd_level_desc> d;
        for (auto &l : m->mesh.levels) d.push_back(l.desc());
        *out = build_solver(d.data(), static_cast<int>(d.size()), m->mesh.mesh_variant, device).release();
    });
}
void mgcfd_destroy(mgcfd_solver *s) { delete s; }

int mgcfd_set_option(mgcfd_solver *s, int option, int value)
{
    REQUIRE(s);
    return guarded([&] {
        switch (option) {
            case MGCFD_OPT_EXACT: s->opt_exact = value != 0; s->upload_optional_plans(); break;
            case MGCFD_OPT_TIMING:
                if (value < 0 || value > 4) throw std::invalid_argument("MGCFD_OPT_TIMING is 0 ... 4");
                s->use_device(); s->fold_events(); s->opt_timing = value == 3 ? 2 : value; s->timing_stride = value == 3 ? 1 : (value == 4 ? 32 : 8);
                if (const char *e = std::getenv("MGCFD_TIMING_STRIDE")) if (std::atoi(e) > 0 && value != 3) s->timing_stride = std::atoi(e);
                break;
            case MGCFD_OPT_INDIRECT_RW: s->opt_indirect_rw = value != 0; break;
            case MGCFD_OPT_CHECK_INVALID: s->opt_check = value != 0; break;
            case MGCFD_OPT_FLUX_VARIANT: s->opt_variant = value; s->upload_optional_plans(); break;
            case MGCFD_OPT_FUSE_UPDATE: s->opt_fuse = value != 0; break;
            case MGCFD_OPT_GRAPH: s->opt_graph = value != 0; break;
            case MGCFD_OPT_RANK_SPLIT: s->opt_rank_split = value < 0 ? 0 : (value > 2 ? 2 : value); break;
            default: throw std::invalid_argument("unknown option");
        }
    });
}
int mgcfd_level_has_half_rows(const mgcfd_solver *s, int level, int *yes)
{
    REQUIRE(s); REQUIRE(yes);
    if (level < 0 || level >= static_cast<int>(s->L.size())) { g_last_error = "level out of range"; return MGCFD_ERR_ARG; }
    *yes = s->L[static_cast<size_t>(level)].dp.half;
    return MGCFD_OK;
}
int mgcfd_level_has_order_free(const mgcfd_solver *s, int level, int *yes)
{
    REQUIRE(s); REQUIRE(yes);
    if (level < 0 || level >= static_cast<int>(s->L.size())) { g_last_error = "level out of range"; return MGCFD_ERR_ARG; }
    *yes = s->L[static_cast<size_t>(level)].dp.free_rows;
    return MGCFD_OK;
}
int mgcfd_level_has_edge_once(const mgcfd_solver *s, int level, int *yes)
{
    REQUIRE(s); REQUIRE(yes);
    if (level < 0 || level >= static_cast<int>(s->L.size())) { g_last_error = "level out of range"; return MGCFD_ERR_ARG; }
    *yes = s->L[static_cast<size_t>(level)].dp.edge_once;
    return MGCFD_OK;
}
int mgcfd_level_tiling(const mgcfd_solver *s, int level, int64_t out[10])
{
    REQUIRE(s); REQUIRE(out);
    if (level < 0 || level >= static_cast<int>(s->L.size())) { g_last_error = "level out of range"; return MGCFD_ERR_ARG; }
    const LevelPlan &P = s->L[static_cast<size_t>(level)].plan;
    out[0] = P.n_tiles;
    out[1] = P.halo_total;
    out[2] = P.halo_max;
    out[3] = kTileCap - kTile;
    out[4] = P.halo_overflow_refs;
    out[5] = P.n_internal_entries;
    out[6] = P.pad_entries;
    out[7] = P.ordered_by_boxes ? 1 : 0;
    out[8] = P.has_tail ? P.tail_total : 0;
    out[9] = 0;
    for (int32_t r : P.rows_main) out[9] += r;
    return MGCFD_OK;
}
int mgcfd_get_option(const mgcfd_solver *s, int option, int *value)
{
    REQUIRE(s); REQUIRE(value);
    switch (option) {
        case MGCFD_OPT_EXACT: *value = s->opt_exact; break;
        case MGCFD_OPT_TIMING: *value = s->opt_timing; break;
        case MGCFD_OPT_INDIRECT_RW: *value = s->opt_indirect_rw; break;
        case MGCFD_OPT_CHECK_INVALID: *value = s->opt_check; break;
        case MGCFD_OPT_FLUX_VARIANT: *value = s->opt_variant; break;
        case MGCFD_OPT_FUSE_UPDATE: *value = s->opt_fuse; break;
        case MGCFD_OPT_GRAPH: *value = s->opt_graph; break;
        case MGCFD_OPT_RANK_SPLIT: *value = s->opt_rank_split; break;
        default: g_last_error = "unknown option"; return MGCFD_ERR_ARG;
    }
    return MGCFD_OK;
}
int mgcfd_set_stream(mgcfd_solver *s, void *hip_stream)
{
    REQUIRE(s);
    return guarded([&] {
        s->use_device();
        s->fold_events();
        HIP_CHECK(hipStreamSynchronize(s->stream));
        s->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : s->own_stream;
    });
}
int mgcfd_synchronize(mgcfd_solver *s)
{
    REQUIRE(s);
    return guarded([&] { s->use_device(); HIP_CHECK(hipStreamSynchronize(s->stream)); HIP_CHECK(hipGetLastError()); fail_on_ipc_timeouts(s); });
}
int mgcfd_num_levels(const mgcfd_solver *s) { return s ? static_cast<int>(s->L.size()) : 0; }
int64_t mgcfd_level_nel(const mgcfd_solver *s, int l)
{ return (s && l >= 0 && l < static_cast<int>(s->L.size())) ? s->L[static_cast<size_t>(l)].info.nel : -1; }
int64_t mgcfd_level_num_internal_edges(const mgcfd_solver *s, int l)
{ return (s && l >= 0 && l < static_cast<int>(s->L.size())) ? s->L[static_cast<size_t>(l)].info.n_internal : -1; }
int mgcfd_get_far_field(const mgcfd_solver *s, double *out17)
{
    REQUIRE(s); REQUIRE(out17);
    std::memcpy(out17, s->ff17, sizeof(double) * 17);
    return MGCFD_OK;
}

// ---- kernel-granular operations ----
#define OP(body) REQUIRE(s); return guarded([&] { s->use_device(); body; HIP_CHECK(hipGetLastError()); })   /* (a launch that failed must not come back as MGCFD_OK) */
int mgcfd_copy_old_variables(mgcfd_solver *s, int level) { OP(s->op_copy_old(level)); }
int mgcfd_compute_step_factor(mgcfd_solver *s, int level) { OP(s->op_step_factor(level)); }
int mgcfd_compute_flux_edge(mgcfd_solver *s, int level) { OP(s->op_flux(level, 1)); }
int mgcfd_compute_boundary_flux_edge(mgcfd_solver *s, int level) { OP(s->op_flux(level, 2)); }
int mgcfd_compute_wall_flux_edge(mgcfd_solver *s, int level) { OP(s->op_flux(level, 4)); }
int mgcfd_compute_fluxes(mgcfd_solver *s, int level) { OP(s->op_flux(level, 7)); }
int mgcfd_time_step(mgcfd_solver *s, int level, int j) { OP(s->op_time_step(level, j)); }
int mgcfd_zero_fluxes(mgcfd_solver *s, int level) { OP(s->op_zero_fluxes(level)); }
int mgcfd_indirect_rw(mgcfd_solver *s, int level) { OP(s->op_indirect_rw(level)); }
int mgcfd_residual(mgcfd_solver *s, int level) { OP(s->op_residual(level)); }
int mgcfd_restrict(mgcfd_solver *s, int fine_level) { OP(s->op_restrict(fine_level)); }
int mgcfd_prolong(mgcfd_solver *s, int fine_level) { OP(s->op_prolong(fine_level)); }
int mgcfd_step_factor_local(mgcfd_solver *s, int level)
{
    OP({
        if (s->mesh_variant == MGCFD_MESH_FVCORR) throw std::invalid_argument("fvcorr uses a local time step: nothing to reduce");
        s->op_step_factor_local(level, false, true);
        s->level(level).iters[MGCFD_LOOP_COMPUTE_STEP] += s->level(level).info.nel;
    });
}
int mgcfd_step_factor_min_devptr(mgcfd_solver *s, int level, void **devptr)
{ REQUIRE(devptr); OP(*devptr = s->level(level).min_dt); }
int mgcfd_step_factor_apply(mgcfd_solver *s, int level) { OP(s->op_step_factor_apply(level)); }
int mgcfd_residual_sumsq(mgcfd_solver *s, int level, void **devptr)
{ REQUIRE(devptr); OP({ s->op_sumsq(level); *devptr = s->level(level).sumsq; }); }

int mgcfd_calc_rms(mgcfd_solver *s, int level, double *rms)
{
    REQUIRE(s); REQUIRE(rms);
    return guarded([&] {
        s->use_device();
        DeviceLevel &lv = s->level(level);
        s->op_sumsq(level);
        double sum = 0.0;
        HIP_CHECK(hipMemcpyAsync(&sum, lv.sumsq, sizeof(double), hipMemcpyDeviceToHost, s->stream));
        HIP_CHECK(hipStreamSynchronize(s->stream));
        *rms = std::sqrt(sum / double(lv.n_owned));          // (a partitioned level: its owned nodes; ghosts are counted by their owners)
    });
}
int mgcfd_invalid_state_location(const mgcfd_solver *s, int64_t *cell, int *cycle)
{
    REQUIRE(s);
    if (cell) *cell = s->invalid_cell;
    if (cycle) *cycle = s->invalid_cycle;
    return MGCFD_OK;
}
int mgcfd_check_for_invalid_variables(mgcfd_solver *s, int level, int64_t *bad_cell)
{
    REQUIRE(s);
    int code = MGCFD_OK;
    int rc = guarded([&] {
        s->use_device();
        DeviceLevel &lv = s->level(level);
        exact::launch_check_invalid(s->stream, lv.info.nel, lv.dp.stride, lv.q, lv.dp.old_of_new, s->err);
        code = s->read_error(bad_cell);
    });
    if (rc != MGCFD_OK) return rc;
    if (code != MGCFD_OK) g_last_error = "invalid variables detected";
    return code;
}

int mgcfd_pending_invalid_state(mgcfd_solver *s, int64_t *bad_cell)
{
    REQUIRE(s);
    int code = MGCFD_OK;
    int rc = guarded([&] { s->use_device(); code = s->read_error(bad_cell); });
    if (rc != MGCFD_OK) return rc;
    if (code != MGCFD_OK) g_last_error = "invalid variables detected";
    return code;
}

// One smoothing sweep = the per-level body of the reference's cycle loop
// (src/euler3d_cpu_double.cpp:383-508): copy, step factor, RK x (fluxes, time_step), residual.
// Same operations, fewer passes over memory: the copy rides on the step-factor kernel, the
// "/ volume" half of the global time step on the first time_step, the residual on the last, and
// the three edge classes share one flux launch.
static void smooth_once(mgcfd_solver *s, int level)
{
    DeviceLevel &lv = s->level(level);
    if (s->opt_fuse && !(s->variant_for(lv) & 4) && !s->opt_indirect_rw && s->opt_timing != 1 && lv.fluxes_zero) {
        // Fused stages: flux + time_step in one launch each.  No copy<double>(old_variables, variables)
        // (:383): the sweep's start state stays where it is and BECOMES old_variables; the stages run
        // variables -> q_alt -> (the former old_variables buffer) -> q_alt, and the three buffers
        // change roles at the end (DeviceLevel::rot).
        const bool global_dt = s->mesh_variant != MGCFD_MESH_FVCORR;
        bool apply_pending = global_dt;
        lv.residuals_stale = false;                // (an unwritten residual of the previous sweep: this sweep's replaces it)
        if (lv.min_ahead) {
            // the launch that produced `variables` already did compute_step_factor's work on them
            // (:388-395): the per-workgroup minima are in partial_min, or (fvcorr) the factors in sf_alt
            if (!global_dt) { lv.sf_par ^= 1; lv.apply_sf(); }
            lv.iters[MGCFD_LOOP_COMPUTE_STEP] += lv.info.nel;
        } else {
            apply_pending = s->op_step_factor(level, true, false);
        }
        // single level: the next sweep starts from this sweep's result, let the last stage look ahead
        const bool look_ahead = s->L.size() == 1 && lv.n_owned == lv.info.nel;   // (a partitioned level's ghosts are stale)
        double *const start = lv.q, *const b1 = lv.q_alt, *const b2 = lv.old_variables;
        // Every workgroup of the first stage takes the minimum over ALL per-workgroup partial minima: fine for a
        // thousand tiles (11 MB of L2 reads at the M6 size), quadratic beyond — large levels reduce them once.
        int apply = apply_pending ? 1 : 0;
        if (apply_pending && lv.plan.n_tiles > 2048) {
            exact::launch_min_reduce(s->stream, lv.info.nel, lv.partial_min, lv.min_dt);
            apply = 2;
        }
        const bool sumsq = lv.want_sumsq && lv.n_owned == lv.info.nel;
        mgcfd_solver::Timed group(s, level, MGCFD_LOOP_FLUX, true, s->opt_timing == 2 ? MGCFD_RK : 1);
        s->op_fused_stage(level, 0, start, b1, apply, false, start);
        s->op_fused_stage(level, 1, b1, b2, 0, false, start);
        // residual (:508) = this stage's result - the sweep's start state.  On a single-level run nothing reads it before
        // the next sweep overwrites it, so the stage leaves the 40 B per node unwritten (its squares still go into the
        // RMS partials) and settle_residuals writes it if anybody asks: both operands stay where they are.
        const bool lazy_res = s->L.size() == 1 && s->opt_lazy_residual;
        s->op_fused_stage(level, 2, b2, b1, 0, true, start, look_ahead, sumsq, nullptr, nullptr, 0, true, nullptr, 0, lazy_res);
        lv.have_sumsq = sumsq;
        lv.rot = (lv.rot + 1) % 3;                 // variables = b1, q_alt = b2, old_variables = start
        lv.apply_rot();
        lv.min_ahead = look_ahead;
        lv.residuals_stale = lazy_res;
        return;
    }
    const bool apply_pending = s->op_step_factor(level, true);     // :383 + :388-395
    int apply0 = apply_pending ? 1 : 0;
    if (apply_pending && lv.plan.n_tiles > 2048) {                 // large level: reduce the partial minima once (see above)
        exact::launch_min_reduce(s->stream, lv.info.nel, lv.partial_min, lv.min_dt);
        apply0 = 2;
    }
    for (int j = 0; j < MGCFD_RK; j++) {                           // :397-506
        s->op_flux(level, 7);
        // the indirect_rw probe reads fluxes[] right after, so zero for real when it is on
        s->op_time_step(level, j, j == 0 ? apply0 : 0, j == MGCFD_RK - 1, !s->opt_indirect_rw);   // + :508 on the last stage
        if (s->opt_indirect_rw && (j == 0 || !s->probe_first_stage_only)) { s->op_indirect_rw(level); s->op_zero_fluxes(level); }
    }
}

// One smoothing sweep.  With MGCFD_OPT_GRAPH the three (or four) launches of a fused sweep are captured once
// per (level, options, buffer rotation, look-ahead state) into a hipGraph and replayed — one host call per
// sweep; by default they are launched directly, which is faster here.  Sweeps that are being
// timed per kernel (OPT_TIMING) always run eagerly.
static void run_sweep(mgcfd_solver *s, int level)
{
    DeviceLevel &lv = s->level(level);
    const int64_t n = s->sweep_counter++;
    bool timed = s->opt_timing == 1 || (s->opt_timing == 2 && (n % s->timing_stride) == 0);     // (stride 1 when MGCFD_OPT_TIMING was set to 3)
    if (s->opt_timing == 4) {
        // Per-loop times by ATTRIBUTION (src/Monitoring/timer.cpp:58-195 wants a time per loop; bracketing every loop means
        // un-fusing every stage: 0.80 ms per cycle against 0.29).  Every Nth sweep of a level runs one launch per loop under
        // events — flux, compute_step, time_step and the indirect_rw probe as the reference's -DTIME build brackets them —
        // all others run fused and un-bracketed; mgcfd_get_loop_times apportions the batches' measured GPU time to the
        // loops by the sampled sweeps' ratios.  Same results either way (the fused stages are the same operations).
        const bool sampled = (lv.att_sweeps % s->timing_stride) == 0;
        // (the indirect_rw probe — not part of the solution — runs in every fourth sampled sweep only; LoopNumIters books it per
        //  stage, as the reference's loop structure has it, and its time is extrapolated at the measured rate)
        const bool probe = sampled && s->opt_indirect_rw && (lv.att_sweeps_sampled % 4) == 0;
        lv.att_sweeps++;
        if (sampled) lv.att_sweeps_sampled++;
        if (probe) lv.att_calls_sampled[MGCFD_LOOP_INDIRECT_RW]++;
        const int keep_t = s->opt_timing, keep_p = s->opt_indirect_rw;
        s->opt_timing = sampled ? 1 : 0;
        // (... and there behind the first Runge-Kutta stage only: the three stages' probes move the same bytes)
        if (keep_p) lv.iters[MGCFD_LOOP_INDIRECT_RW] += int64_t(probe ? MGCFD_RK - 1 : MGCFD_RK) * lv.info.n_internal;
        if (!probe) s->opt_indirect_rw = 0;
        s->probe_first_stage_only = true;
        try { smooth_once(s, level); } catch (...) { s->opt_timing = keep_t; s->opt_indirect_rw = keep_p; s->probe_first_stage_only = false; throw; }
        s->opt_timing = keep_t; s->opt_indirect_rw = keep_p; s->probe_first_stage_only = false;
        return;
    }
    // (only the fused launches are replayed: the unfused ones — the two-phase flux variant — leave host-side flags
    //  behind, fluxes_stale, that a replay would not set)
    const bool graphable = s->opt_graph && s->opt_fuse && !(s->variant_for(lv) & 4) && !s->opt_indirect_rw && !timed && lv.fluxes_zero && !lv.fluxes_stale;
    if (!graphable) {
        const int keep = s->opt_timing;
        if (!timed) s->opt_timing = 0;
        smooth_once(s, level);
        s->opt_timing = keep;
        return;
    }
    const uint64_t key = (uint64_t(level) << 32) | (uint64_t(lv.want_sumsq) << 28) | (uint64_t(lv.sf_par) << 27) | (uint64_t(lv.rot) << 25) | (uint64_t(lv.min_ahead) << 24) | (uint64_t(s->opt_exact) << 16) | (uint64_t(s->opt_check) << 8) | uint64_t(s->opt_variant & 0xFF);
    auto it = s->sweep_graphs.find(key);
    if (it == s->sweep_graphs.end()) {
        mgcfd_solver::SweepGraph g;
        int64_t before[MGCFD_NUM_LOOPS];
        std::memcpy(before, lv.iters, sizeof(before));
        const int keep = s->opt_timing;
        s->opt_timing = 0;
        hipGraph_t graph = nullptr;
        HIP_CHECK(hipStreamBeginCapture(s->stream, hipStreamCaptureModeRelaxed));
        try {
            smooth_once(s, level);
        } catch (...) {
            (void)hipStreamEndCapture(s->stream, &graph);
            if (graph) (void)hipGraphDestroy(graph);
            s->opt_timing = keep;
            throw;
        }
        s->opt_timing = keep;
        HIP_CHECK(hipStreamEndCapture(s->stream, &graph));
        HIP_CHECK(hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0));
        HIP_CHECK(hipGraphDestroy(graph));
        for (int k = 0; k < MGCFD_NUM_LOOPS; k++) { g.iters[k] = lv.iters[k] - before[k]; lv.iters[k] = before[k]; }
        g.ahead_after = lv.min_ahead;
        g.rot_after = lv.rot;
        g.sf_par_after = lv.sf_par;
        g.sumsq_after = lv.have_sumsq;
        g.res_stale_after = lv.residuals_stale;
        it = s->sweep_graphs.emplace(key, g).first;
    }
    HIP_CHECK(hipGraphLaunch(it->second.exec, s->stream));
    lv.min_ahead = it->second.ahead_after;
    lv.rot = it->second.rot_after;
    lv.apply_rot();
    lv.sf_par = it->second.sf_par_after;
    lv.apply_sf();
    lv.have_sumsq = it->second.sumsq_after;
    lv.residuals_stale = it->second.res_stale_after;
    for (int k = 0; k < MGCFD_NUM_LOOPS; k++) lv.iters[k] += it->second.iters[k];
}

// The same sweep split around the one collective a multi-GPU run needs (see mgcfd.h).
static int sweep_begin_impl(mgcfd_solver *s, int level, bool scalar)
{
    OP({
        DeviceLevel &lv = s->level(level);
        if (!lv.fluxes_zero) throw std::invalid_argument("sweep_begin needs zero fluxes (as after time_step)");
        lv.sweep_flux0_done = false;
        // As in smooth_once: no copy (the start state stays in `variables` and becomes old_variables
        // when sweep_end rotates the buffers), and no step-factor kernel when the launch that produced
        // `variables` already left the minima behind.
        const bool global_dt = s->mesh_variant != MGCFD_MESH_FVCORR;
        if (lv.min_ahead) {
            if (!global_dt) { lv.sf_par ^= 1; lv.apply_sf(); }
            lv.iters[MGCFD_LOOP_COMPUTE_STEP] += lv.info.nel;
        } else {
            s->op_step_factor(level, true, false);
        }
        if (global_dt && scalar) exact::launch_min_reduce(s->stream, lv.info.nel, lv.partial_min, lv.min_dt);
    });
}
int mgcfd_sweep_flux0(mgcfd_solver *s, int level)
{
    OP({
        DeviceLevel &lv = s->level(level);
        if (!lv.fluxes_zero) throw std::invalid_argument("sweep_flux0 must follow sweep_begin");
        s->op_flux(level, 7);                      // stage-0 fluxes: independent of the time step
        lv.sweep_flux0_done = true;
    });
}
static int sweep_end_impl(mgcfd_solver *s, int level, bool scalar)
{
    OP({
        DeviceLevel &lv = s->level(level);
        const bool global_dt = s->mesh_variant != MGCFD_MESH_FVCORR;
        const int apply = global_dt ? (scalar ? 2 : 1) : 0;
        const bool look_ahead = s->L.size() == 1 && lv.n_owned == lv.info.nel;
        double *const start = lv.q;
        double *const b1 = lv.q_alt;
        double *const b2 = lv.old_variables;
        if (lv.sweep_flux0_done && global_dt && lv.dp.vin_ok && !((s->variant_for(lv) & 2) && lv.dp.edge_once)) {
            // the second stage applies the first stage's time_step (on the fluxes of sweep_flux0) to its own
            // input while it stages it: no separate time_step launch, the first stage's result never touches memory
            s->op_fused_stage(level, 1, start, b2, apply, false, start, false, false, lv.fluxes);
            lv.iters[MGCFD_LOOP_TIME_STEP] += lv.info.nel;      // the first stage's time_step
            lv.fluxes_zero = true;                               // ... which leaves fluxes[] logically zero
            lv.fluxes_stale = true;
            lv.sweep_flux0_done = false;
        } else {
            if (lv.sweep_flux0_done) {
                s->op_time_step(level, 0, apply, false, true, start, b1);   // time_step on the fluxes of sweep_flux0
                lv.sweep_flux0_done = false;
            } else {
                s->settle_fluxes(lv);
                s->op_fused_stage(level, 0, start, b1, apply, false, start);
            }
            s->op_fused_stage(level, 1, b1, b2, 0, false, start);
        }
        s->op_fused_stage(level, 2, b2, b1, 0, true, start, look_ahead);
        lv.rot = (lv.rot + 1) % 3;                 // variables = b1, q_alt = b2, old_variables = start
        lv.apply_rot();
        lv.min_ahead = look_ahead;
    });
}

int mgcfd_sweep_stage(mgcfd_solver *s, int level, int j, int partials)
{
    OP({
        DeviceLevel &lv = s->level(level);
        if (j != lv.stage_next) throw std::invalid_argument("mgcfd_sweep_stage: stages must run in order 0, 1, 2 after mgcfd_sweep_begin");
        const bool global_dt = s->mesh_variant != MGCFD_MESH_FVCORR;
        double *const start = lv.q;
        double *const b1 = lv.q_alt;
        double *const b2 = lv.old_variables;
        if (j == 0) {
            if (!lv.fluxes_zero) throw std::invalid_argument("mgcfd_sweep_stage needs zero fluxes (as after time_step)");
            s->settle_fluxes(lv);
            s->op_fused_stage(level, 0, start, b1, global_dt ? (partials ? 1 : 2) : 0, false, start);
            lv.stage_out = b1;
            lv.stage_next = 1;
        } else if (j == 1) {
            s->op_fused_stage(level, 1, b1, b2, 0, false, start);
            lv.stage_out = b2;
            lv.stage_next = 2;
        } else {
            // (no look-ahead: on a partitioned level the ghosts of the new state are stale until the exchange)
            s->op_fused_stage(level, 2, b2, b1, 0, true, start, false);
            lv.rot = (lv.rot + 1) % 3;             // variables = b1, q_alt = b2, old_variables = start
            lv.apply_rot();
            lv.stage_out = lv.q;                   // == b1
            lv.stage_next = 0;
            lv.min_ahead = false;
        }
    });
}
int mgcfd_sweep_begin(mgcfd_solver *s, int level) { return sweep_begin_impl(s, level, true); }
int mgcfd_sweep_end(mgcfd_solver *s, int level) { return sweep_end_impl(s, level, true); }
int mgcfd_sweep_begin_partials(mgcfd_solver *s, int level) { return sweep_begin_impl(s, level, false); }
int mgcfd_sweep_end_partials(mgcfd_solver *s, int level) { return sweep_end_impl(s, level, false); }
int mgcfd_step_factor_partials_devptr(mgcfd_solver *s, int level, void **devptr, int *count)
{
    REQUIRE(devptr); REQUIRE(count);
    OP({ DeviceLevel &lv = s->level(level); *devptr = lv.partial_min; *count = static_cast<int>((lv.info.nel + 255) / 256); });
}

int mgcfd_smooth(mgcfd_solver *s, int level, int sweeps)
{
    OP({
        s->level(level);
        for (int k = 0; k < sweeps; k++) run_sweep(s, level);
    });
}

// ---- cycle driver: src/euler3d_cpu_double.cpp:371-694 ----
// One multigrid cycle of the reference's state machine, unrolled: sweeps on levels
// 0,1,..,n-1,n-2,..,1 with restrictions on the way up and prolongations on the way down
// (src/euler3d_cpu_double.cpp:371-694); the level-0 sum of squares is appended to the rms ring.
static void cycle_once(mgcfd_solver *s, bool capturing)
{
    const int n = static_cast<int>(s->L.size());
    auto sweep = [&](int l) { if (capturing) smooth_once(s, l); else run_sweep(s, l); };
    SumTask rms_task;
    bool rms_pending = false;
    // (MGCFD_OPT_TIMING = 4: every Nth restriction / prolongation of a level pair is bracketed with events, see run_sweep)
    auto transfer = [&](bool restriction, int fine, const SumTask *task) {
        const int keep = s->opt_timing;
        if (keep == 4) {
            DeviceLevel &book = s->L[static_cast<size_t>(restriction ? fine + 1 : fine)];     // (the reference books restrict on the coarse level)
            const int loop = restriction ? MGCFD_LOOP_RESTRICT : MGCFD_LOOP_PROLONG;
            const bool sampled = (book.att_calls[loop] % s->timing_stride) == 0;
            book.att_calls[loop]++;
            if (sampled) book.att_calls_sampled[loop]++;
            s->opt_timing = sampled ? 1 : 0;
        }
        try { if (restriction) s->op_restrict(fine, task); else s->op_prolong(fine); } catch (...) { s->opt_timing = keep; throw; }
        s->opt_timing = keep;
    };
    for (int l = 0; l < n; l++) {
        if (l == 0) { s->L[0].want_sumsq = true; s->L[0].have_sumsq = false; }
        sweep(l);                                                          // :383-508
        if (l == 0) {                                                      // :509-512
            DeviceLevel &l0 = s->L[0];
            l0.want_sumsq = false;
            if (l0.have_sumsq && n > 1) {
                // the last stage left per-tile sums of squares: the restriction that follows adds them up and appends
                rms_task = SumTask{l0.tile_sumsq, static_cast<int>((l0.info.nel + 255) / 256), l0.sumsq, s->rms_ring, s->rms_count,
                                   mgcfd_solver::kRmsRing};
                rms_pending = true;
            } else if (l0.have_sumsq) {
                // ... single level: one small launch does
                exact::launch_sum_partials_append(s->stream, static_cast<int>((l0.info.nel + 255) / 256), l0.tile_sumsq, l0.sumsq,
                                                  s->rms_ring, s->rms_count, mgcfd_solver::kRmsRing);
            } else {
                s->op_sumsq(0);
                exact::launch_append_scalar(s->stream, l0.sumsq, s->rms_ring, s->rms_count, mgcfd_solver::kRmsRing);
            }
        }
        if (l + 1 < n) { transfer(true, l, (l == 0 && rms_pending) ? &rms_task : nullptr); rms_pending = false; }   // :527-559
    }
    for (int l = n - 2; l >= 0; l--) {
        transfer(false, l, nullptr);                                       // :560-688
        if (l > 0) sweep(l);
    }
}

int mgcfd_run_cycles(mgcfd_solver *s, int cycles, double *rms_out)
{
    REQUIRE(s);
    int code = MGCFD_OK, failed_cycle = -1, seq_before = 0, seq_per_cycle = 0;
    int rc = guarded([&] {
        s->use_device();
        if (!s->rms_ring) {
            s->rms_ring = dev_alloc<double>(mgcfd_solver::kRmsRing);
            s->rms_count = dev_alloc<int>(1);
        }
        const size_t nl = s->L.size();
        std::vector<double> sums;
        sums.reserve(static_cast<size_t>(cycles > 0 ? cycles : 0));
        seq_before = s->check_seq;                  // checked launches of earlier calls nobody has read back yet
        for (int done = 0; done < cycles;) {
            const int chunk = std::min(cycles - done, mgcfd_solver::kRmsRing);
            HIP_CHECK(hipMemsetAsync(s->rms_count, 0, sizeof(int), s->stream));
            hipEvent_t att0 = nullptr, att1 = nullptr;
            if (s->opt_timing == 4) { att0 = s->get_event(); att1 = s->get_event(); HIP_CHECK(hipEventRecord(att0, s->stream)); }
            bool graphable = s->opt_graph && s->opt_fuse && !s->opt_indirect_rw && s->opt_timing == 0;
            for (auto &lv : s->L) graphable = graphable && lv.fluxes_zero && !lv.fluxes_stale && !(s->variant_for(lv) & 4);
            graphable = graphable && nl <= 8;               // the graph key holds 8 levels' buffer rotations
            if (graphable) {
                // the whole cycle — every sweep and transfer of every level — as ONE graph replay.
                // The launch sequence depends on which levels enter with their step-factor minima
                // already computed (min_ahead), so that is part of the key and looked up per cycle.
                for (int c = 0; c < chunk; c++) {
                    uint64_t key = (uint64_t(s->opt_exact) << 16) | (uint64_t(s->opt_check) << 8) | uint64_t(s->opt_variant & 0xFF);
                    for (size_t l = 0; l < nl && l < 8; l++) key |= (uint64_t(s->L[l].min_ahead) << (24 + l)) | (uint64_t(s->L[l].rot) << (32 + 2 * l)) | (uint64_t(s->L[l].sf_par) << (48 + l));
                    auto it = s->cycle_graphs.find(key);
                    if (it == s->cycle_graphs.end()) {
                        mgcfd_solver::CycleGraph g;
                        std::vector<std::vector<int64_t>> before(nl);
                        std::vector<bool> ahead_before, stale_before;
                        std::vector<int> rot_before, sf_before;
                        for (size_t l = 0; l < nl; l++) {
                            before[l].assign(s->L[l].iters, s->L[l].iters + MGCFD_NUM_LOOPS);
                            ahead_before.push_back(s->L[l].min_ahead);
                            stale_before.push_back(s->L[l].residuals_stale);
                            rot_before.push_back(s->L[l].rot);
                            sf_before.push_back(s->L[l].sf_par);
                        }
                        hipGraph_t graph = nullptr;
                        HIP_CHECK(hipStreamBeginCapture(s->stream, hipStreamCaptureModeRelaxed));
                        try {
                            cycle_once(s, true);
                        } catch (...) {
                            (void)hipStreamEndCapture(s->stream, &graph);
                            if (graph) (void)hipGraphDestroy(graph);
                            throw;
                        }
                        HIP_CHECK(hipStreamEndCapture(s->stream, &graph));
                        HIP_CHECK(hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0));
                        HIP_CHECK(hipGraphDestroy(graph));
                        g.iters.resize(nl);
                        for (size_t l = 0; l < nl; l++) {
                            for (int k = 0; k < MGCFD_NUM_LOOPS; k++) {
                                g.iters[l].push_back(s->L[l].iters[k] - before[l][static_cast<size_t>(k)]);
                                s->L[l].iters[k] = before[l][static_cast<size_t>(k)];
                            }
                            g.ahead_after.push_back(s->L[l].min_ahead);
                            g.rot_after.push_back(s->L[l].rot);
                            g.sf_par_after.push_back(s->L[l].sf_par);
                            g.res_stale_after.push_back(s->L[l].residuals_stale);
                            s->L[l].residuals_stale = stale_before[l];
                            s->L[l].min_ahead = ahead_before[l];          // nothing ran yet: capture only recorded
                            s->L[l].rot = rot_before[l];
                            s->L[l].apply_rot();
                            s->L[l].sf_par = sf_before[l];
                            s->L[l].apply_sf();
                        }
                        it = s->cycle_graphs.emplace(key, std::move(g)).first;
                    }
                    HIP_CHECK(hipGraphLaunch(it->second.exec, s->stream));
                    for (size_t l = 0; l < nl; l++) {
                        for (int k = 0; k < MGCFD_NUM_LOOPS; k++) s->L[l].iters[k] += it->second.iters[l][static_cast<size_t>(k)];
                        s->L[l].min_ahead = it->second.ahead_after[l];
                        s->L[l].residuals_stale = it->second.res_stale_after[l];   // (a single level's cycle leaves its residual unwritten)
                        s->L[l].rot = it->second.rot_after[l];
                        s->L[l].apply_rot();
                        s->L[l].sf_par = it->second.sf_par_after[l];
                        s->L[l].apply_sf();
                    }
                }
            } else {
                for (int c = 0; c < chunk; c++) {
                    cycle_once(s, false);
                    if (c == 0) seq_per_cycle = s->check_seq - seq_before;
                }
            }
            if (att1) HIP_CHECK(hipEventRecord(att1, s->stream));
            const size_t at = sums.size();
            sums.resize(at + static_cast<size_t>(chunk));
            HIP_CHECK(hipMemcpyAsync(sums.data() + at, s->rms_ring, sizeof(double) * chunk, hipMemcpyDeviceToHost, s->stream));
            int seq = 0;
            code = s->read_error(nullptr, &seq);                           // synchronises
            if (att1) {
                float ms = 0.f;
                HIP_CHECK(hipEventElapsedTime(&ms, att0, att1));
                s->att_total += double(ms) * 1e-3;
                s->free_events.push_back(att0); s->free_events.push_back(att1);
            }
            if (code != MGCFD_OK) {
                // the reference exits inside the failing time_step: stop here, and say in which cycle it was when
                // the launches were numbered one by one (a replayed graph repeats its numbers)
                if (seq_per_cycle > 0 && seq > seq_before)
                    failed_cycle = done + std::min(chunk - 1, (seq - 1 - seq_before) / seq_per_cycle);
                done += chunk;
                break;
            }
            done += chunk;
            seq_before = 0;
        }
        if (rms_out) {
            const double nan = std::numeric_limits<double>::quiet_NaN();
            for (int c = 0; c < cycles; c++)
                rms_out[c] = (c < static_cast<int>(sums.size()) && (failed_cycle < 0 || c < failed_cycle))
                                 ? std::sqrt(sums[static_cast<size_t>(c)] / double(s->L[0].n_owned)) : nan;
        }
        HIP_CHECK(hipGetLastError());
    });
    if (rc != MGCFD_OK) return rc;
    if (code != MGCFD_OK) {
        s->invalid_cycle = failed_cycle;
        g_last_error = std::string("check_for_invalid_variables: ") +
                       (code == MGCFD_ERR_NAN ? "NaN/Inf" : code == MGCFD_ERR_NEG_DENSITY ? "negative density" : "negative density*energy") +
                       " at cell " + std::to_string(s->invalid_cell) +
                       (failed_cycle >= 0 ? " in cycle " + std::to_string(failed_cycle + 1) : std::string(" during the cycles"));
    }
    return code;
}

// ---- state access ----
static double *array_ptr(DeviceLevel &lv, int which, int *ncols)
{
    *ncols = 5;
    switch (which) {
        case MGCFD_ARR_VARIABLES: return lv.q;
        case MGCFD_ARR_OLD_VARIABLES: return lv.old_variables;
        case MGCFD_ARR_FLUXES: return lv.fluxes;
        case MGCFD_ARR_RESIDUALS: return lv.residuals;
        case MGCFD_ARR_STEP_FACTORS: *ncols = 1; return lv.step_factors;
        case MGCFD_ARR_VOLUMES: *ncols = 1; return lv.volumes;
        case MGCFD_ARR_STAGE:
            if (!lv.stage_out) throw std::invalid_argument("MGCFD_ARR_STAGE: no mgcfd_sweep_stage has run on this level");
            return lv.stage_out;
        default: throw std::invalid_argument("unknown array id");
    }
}
int mgcfd_get_array(mgcfd_solver *s, int level, int which, double *out)
{
    REQUIRE(s); REQUIRE(out);
    return guarded([&] {
        s->use_device();
        DeviceLevel &lv = s->level(level);
        int nc = 0;
        double *src = (s->settle_residuals(lv), array_ptr(lv, which, &nc));
        if (which == MGCFD_ARR_FLUXES) s->settle_fluxes(lv);
        const int64_t stride = lv.dp.stride;
        std::vector<double> tmp(static_cast<size_t>(stride) * nc);
        HIP_CHECK(hipMemcpyAsync(tmp.data(), src, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream));
        HIP_CHECK(hipStreamSynchronize(s->stream));
        fail_on_ipc_timeouts(s);                             // (an array computed from stale ghosts is not handed out as if it were good)
        for (int64_t n = 0; n < lv.info.nel; n++) {          // device: [field][new id]  ->  caller: [old id][field]
            const int64_t o = lv.plan.old_of_new[static_cast<size_t>(n)];
            for (int c = 0; c < nc; c++) out[o * nc + c] = tmp[static_cast<size_t>(c * stride + n)];
        }
    });
}
int mgcfd_set_array(mgcfd_solver *s, int level, int which, const double *in)
{
    REQUIRE(s); REQUIRE(in);
    return guarded([&] {
        s->use_device();
        DeviceLevel &lv = s->level(level);
        int nc = 0;
        double *dst = (s->settle_residuals(lv), array_ptr(lv, which, &nc));
        if (which == MGCFD_ARR_VOLUMES) throw std::invalid_argument("volumes are fixed at creation");
        const int64_t stride = lv.dp.stride;
        // keep the padded tail of every field as it is on the device (valid numbers)
        std::vector<double> tmp(static_cast<size_t>(stride) * nc);
        HIP_CHECK(hipMemcpyAsync(tmp.data(), dst, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream));
        HIP_CHECK(hipStreamSynchronize(s->stream));
        for (int64_t n = 0; n < lv.info.nel; n++) {
            const int64_t o = lv.plan.old_of_new[static_cast<size_t>(n)];
            for (int c = 0; c < nc; c++) tmp[static_cast<size_t>(c * stride + n)] = in[o * nc + c];
        }
        HIP_CHECK(hipMemcpyAsync(dst, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice, s->stream));
        HIP_CHECK(hipStreamSynchronize(s->stream));
        if (which == MGCFD_ARR_FLUXES) { lv.fluxes_zero = false; lv.fluxes_stale = false; }
        if (which == MGCFD_ARR_VARIABLES) lv.min_ahead = false;
    });
}
int mgcfd_array_devptr(mgcfd_solver *s, int level, int which, void **devptr, int64_t *count)
{
    REQUIRE(s); REQUIRE(devptr); REQUIRE(count);
    return guarded([&] {
        s->use_device();
        DeviceLevel &lv = s->level(level);
        int nc = 0;
        double *p = (s->settle_residuals(lv), array_ptr(lv, which, &nc));
        if (which == MGCFD_ARR_FLUXES) s->settle_fluxes(lv);
        *devptr = p;
        *count = int64_t(nc) * lv.dp.stride;
    });
}
int mgcfd_array_written(mgcfd_solver *s, int level, int which)
{
    REQUIRE(s);
    return guarded([&] {
        DeviceLevel &lv = s->level(level);
        int nc = 0;
        (void)(s->settle_residuals(lv), array_ptr(lv, which, &nc));
        if (which == MGCFD_ARR_VOLUMES) throw std::invalid_argument("volumes are fixed at creation");
        if (which == MGCFD_ARR_FLUXES) { lv.fluxes_zero = false; lv.fluxes_stale = false; }
        if (which == MGCFD_ARR_VARIABLES) lv.min_ahead = false;
    });
}
// One level per rank: take a restricted `variables` array computed by the rank that holds the finer level.
int mgcfd_accept_restricted(mgcfd_solver *s, int fine_level, const void *dev_src)
{
    REQUIRE(s); REQUIRE(dev_src);
    return guarded([&] {
        s->use_device();
        DeviceLevel &fine = s->level(fine_level);
        DeviceLevel &coarse = s->level(fine_level + 1);
        if (!fine.dp.child_ptr) throw std::invalid_argument("level has no coarser level");
        exact::launch_accept_restricted(s->stream, coarse.info.nel, coarse.dp.stride, fine.dp.child_ptr,
                                        static_cast<const double *>(dev_src), coarse.q);
        coarse.min_ahead = false;
    });
}
int mgcfd_get_edges(mgcfd_solver *s, int level, mgcfd_edge *out)
{
    REQUIRE(s); REQUIRE(out);
    return guarded([&] {
        DeviceLevel &lv = s->level(level);
        std::memcpy(out, lv.edges.data(), lv.edges.size() * sizeof(mgcfd_edge));
    });
}

// ---- halo exchange of a partitioned level ----
int mgcfd_halo_plan(mgcfd_solver *s, int level, int64_t n, const int64_t *node_ids, int *plan)
{
    REQUIRE(s); REQUIRE(plan);
    if (n > 0) REQUIRE(node_ids);
    return guarded([&] {
        s->use_device();
        DeviceLevel &lv = s->level(level);
        std::vector<int32_t> ids(static_cast<size_t>(n));
        for (int64_t k = 0; k < n; k++) {
            if (node_ids[k] < 0 || node_ids[k] >= lv.info.nel) throw std::invalid_argument("halo node id out of range");
            ids[static_cast<size_t>(k)] = lv.plan.new_of_old[static_cast<size_t>(node_ids[k])];
        }
        lv.halo_plans.emplace_back(dev_upload(ids), n);
        *plan = static_cast<int>(lv.halo_plans.size()) - 1;
    });
}
static void halo_move(mgcfd_solver *s, int level, int plan, int which, void *dev_buf, bool pack)
{
    s->use_device();
    DeviceLevel &lv = s->level(level);
    if (plan < 0 || plan >= static_cast<int>(lv.halo_plans.size())) throw std::invalid_argument("unknown halo plan");
    int nc = 0;
    double *field = (s->settle_residuals(lv), array_ptr(lv, which, &nc));
    if (nc != 5) throw std::invalid_argument("halo messages carry 5-component node arrays");
    if (which == MGCFD_ARR_FLUXES) { if (pack) s->settle_fluxes(lv); else { lv.fluxes_zero = false; lv.fluxes_stale = false; } }
    if (which == MGCFD_ARR_VARIABLES && !pack) lv.min_ahead = false;
    const auto &hp = lv.halo_plans[static_cast<size_t>(plan)];
    if (hp.second > 0 && !dev_buf) throw std::invalid_argument("null message buffer");
    if (pack) exact::launch_halo_pack(s->stream, hp.second, lv.dp.stride, hp.first, field, static_cast<double *>(dev_buf));
    else exact::launch_halo_unpack(s->stream, hp.second, lv.dp.stride, hp.first, static_cast<const double *>(dev_buf), field);
}
int mgcfd_halo_pack(mgcfd_solver *s, int level, int plan, int which, void *dev_buf)
{ REQUIRE(s); return guarded([&] { halo_move(s, level, plan, which, dev_buf, true); }); }
int mgcfd_halo_unpack(mgcfd_solver *s, int level, int plan, int which, const void *dev_buf)
{ REQUIRE(s); return guarded([&] { halo_move(s, level, plan, which, const_cast<void *>(dev_buf), false); }); }

// ---- monitoring ----
int mgcfd_get_loop_iters(const mgcfd_solver *s, int level, int64_t out[MGCFD_NUM_LOOPS])
{
    REQUIRE(s); REQUIRE(out);
    if (level < 0 || level >= static_cast<int>(s->L.size())) { g_last_error = "level out of range"; return MGCFD_ERR_ARG; }
    std::memcpy(out, s->L[static_cast<size_t>(level)].iters, sizeof(int64_t) * MGCFD_NUM_LOOPS);
    return MGCFD_OK;
}
int mgcfd_get_loop_times(mgcfd_solver *s, int level, double out[MGCFD_NUM_LOOPS])
{
    REQUIRE(s); REQUIRE(out);
    return guarded([&] {
        s->use_device();
        s->fold_events();
        std::memcpy(out, s->level(level).times, sizeof(double) * MGCFD_NUM_LOOPS);
        if (s->opt_timing != 4 || s->att_total <= 0.0) return;
        // attribution (see run_sweep): what the un-bracketed launches of every level and loop would have taken at the sampled
        // rate, scaled so that everything adds up to the GPU time the cycle batches really took
        auto estimate = [&](const DeviceLevel &lv, int k) {
            if (k == MGCFD_LOOP_RESTRICT || k == MGCFD_LOOP_PROLONG)
                return lv.att_calls_sampled[k] > 0 ? lv.times[k] / double(lv.att_calls_sampled[k]) * double(lv.att_calls[k] - lv.att_calls_sampled[k]) : 0.0;
            if (k == MGCFD_LOOP_INDIRECT_RW) return 0.0;        // (the probe does not run in the fused sweeps: none of their time is its)
            return lv.att_sweeps_sampled > 0 ? lv.times[k] / double(lv.att_sweeps_sampled) * double(lv.att_sweeps - lv.att_sweeps_sampled) : 0.0;
        };
        double measured = 0.0, estimated = 0.0;
        for (const DeviceLevel &lv : s->L)
            for (int k = 0; k < MGCFD_NUM_LOOPS; k++) { measured += lv.times[k]; estimated += estimate(lv, k); }
        const double rest = std::max(0.0, s->att_total - measured);
        const double scale = estimated > 0.0 ? rest / estimated : 0.0;
        const DeviceLevel &lv = s->level(level);
        for (int k = 0; k < MGCFD_NUM_LOOPS; k++) out[k] = lv.times[k] + scale * estimate(lv, k);
        // indirect_rw: the measured probes' rate applied to every sweep LoopNumIters books (a rate column: iterations / time
        // stays the probe's measured rate; this time is NOT part of the batches' total)
        if (lv.att_calls_sampled[MGCFD_LOOP_INDIRECT_RW] > 0)
            out[MGCFD_LOOP_INDIRECT_RW] = lv.times[MGCFD_LOOP_INDIRECT_RW] / double(lv.att_calls_sampled[MGCFD_LOOP_INDIRECT_RW]) * double(MGCFD_RK) * double(lv.att_sweeps);
    });
}
int mgcfd_reset_monitoring(mgcfd_solver *s)
{
    REQUIRE(s);
    return guarded([&] {
        s->use_device();
        s->fold_events();
        for (auto &lv : s->L) {
            std::memset(lv.iters, 0, sizeof(lv.iters));
            std::memset(lv.times, 0, sizeof(lv.times));
            lv.flux_time = 0.0;
            lv.flux_launches = 0;
            lv.att_sweeps = lv.att_sweeps_sampled = 0;
            std::memset(lv.att_calls, 0, sizeof(lv.att_calls));
            std::memset(lv.att_calls_sampled, 0, sizeof(lv.att_calls_sampled));
        }
        s->att_total = 0.0;
    });
}
int mgcfd_get_flux_kernel_time(mgcfd_solver *s, int level, double *avg_seconds, int64_t *launches)
{
    REQUIRE(s); REQUIRE(avg_seconds); REQUIRE(launches);
    return guarded([&] {
        s->use_device();
        s->fold_events();
        DeviceLevel &lv = s->level(level);
        *launches = lv.flux_launches;
        *avg_seconds = lv.flux_launches ? lv.flux_time / double(lv.flux_launches) : 0.0;
    });
}

// Diagnostic: time `launches` back-to-back flux launches (all classes, from zero) between two
// hipEvents on the solver's stream; the state is left as after one compute_fluxes call.
int mgcfd_bench_flux(mgcfd_solver *s, int level, int launches, double *avg_seconds)
{
    REQUIRE(s); REQUIRE(avg_seconds);
    return guarded([&] {
        s->use_device();
        DeviceLevel &lv = s->level(level);
        hipEvent_t a = s->get_event(), b = s->get_event();
        const int variant = s->variant_for(lv);
        if ((variant & 4) && !lv.dp.edge_flux)
            lv.dp.edge_flux = dev_alloc<double>(static_cast<size_t>(lv.dp.n_edges_pad) * 5 + 8);
        auto go = [&] {
            if (s->opt_exact) exact::launch_flux(s->stream, lv.dp, lv.q, s->ff, lv.fluxes, 7, 0, variant, nullptr);
            else fast::launch_flux(s->stream, lv.dp, lv.q, s->ff, lv.fluxes, 7, 0, variant, nullptr);
        };
        go();
        HIP_CHECK(hipEventRecord(a, s->stream));
        for (int k = 0; k < launches; k++) go();
        HIP_CHECK(hipEventRecord(b, s->stream));
        HIP_CHECK(hipStreamSynchronize(s->stream));
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, a, b));
        s->free_events.push_back(a);
        s->free_events.push_back(b);
        lv.fluxes_zero = false;
        *avg_seconds = launches > 0 ? double(ms) * 1e-3 / launches : 0.0;
    });
}

// The same for the indirect_rw probe (fluxes += ..., accumulating over the launches): the data-movement ceiling of the
// flux kernel on this level's tiles.
int mgcfd_bench_indirect_rw(mgcfd_solver *s, int level, int launches, double *avg_seconds)
{
    REQUIRE(s); REQUIRE(avg_seconds);
    return guarded([&] {
        s->use_device();
        DeviceLevel &lv = s->level(level);
        s->settle_fluxes(lv);
        hipEvent_t a = s->get_event(), b = s->get_event();
        const int variant = s->variant_for(lv);
        auto go = [&] {
            if (s->opt_exact) exact::launch_indirect_rw(s->stream, lv.dp, lv.q, lv.fluxes, variant);
            else fast::launch_indirect_rw(s->stream, lv.dp, lv.q, lv.fluxes, variant);
        };
        go();
        HIP_CHECK(hipEventRecord(a, s->stream));
        for (int k = 0; k < launches; k++) go();
        HIP_CHECK(hipEventRecord(b, s->stream));
        HIP_CHECK(hipStreamSynchronize(s->stream));
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, a, b));
        s->free_events.push_back(a);
        s->free_events.push_back(b);
        lv.fluxes_zero = false;
        *avg_seconds = launches > 0 ? double(ms) * 1e-3 / launches : 0.0;
    });
}

// The practical ceiling of that launch's data movement: a tile-shaped stream of exactly the bytes SURVEY.md §8d prices
// (40 B per internal edge + 40 B per node read from a scratch buffer of that size, 40 B per node written into `fluxes`), one
// workgroup per tile of the level, nothing dependent and nothing computed (kernels.hip: k_stream_tiles).
int mgcfd_bench_stream_ceiling(mgcfd_solver *s, int level, int launches, double *avg_seconds)
{
    REQUIRE(s); REQUIRE(avg_seconds);
    return guarded([&] {
        s->use_device();
        DeviceLevel &lv = s->level(level);
        s->settle_fluxes(lv);
        const int64_t rd_total = (5 * int64_t(lv.info.n_internal) + 5 * lv.dp.nel + 1) & ~int64_t(1), wr_total = 5 * lv.dp.nel;
        double *src = dev_alloc<double>(static_cast<size_t>(rd_total) + 512);
        HIP_CHECK(hipMemsetAsync(src, 0, (static_cast<size_t>(rd_total) + 512) * sizeof(double), s->stream));
        hipEvent_t a = s->get_event(), b = s->get_event();
        auto go = [&] { exact::launch_stream_tiles(s->stream, lv.dp.n_tiles, src, lv.fluxes, rd_total, wr_total); };
        go();
        HIP_CHECK(hipEventRecord(a, s->stream));
        for (int k = 0; k < launches; k++) go();
        HIP_CHECK(hipEventRecord(b, s->stream));
        HIP_CHECK(hipStreamSynchronize(s->stream));
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, a, b));
        s->free_events.push_back(a);
        s->free_events.push_back(b);
        HIP_CHECK(hipFree(src));
        lv.fluxes_zero = false;
        *avg_seconds = launches > 0 ? double(ms) * 1e-3 / launches : 0.0;
    });
}

} // extern "C"

// ==========================================================================================================
// Multi-GPU in the C++ host (include/mgcfd.h "Multi-GPU in the C++ host").
//
// One level partitioned over ranks: mgcfd_rank_set_halo gives a solver its neighbours and the node lists of the
// messages; mgcfd_rank_sweeps then runs whole smoothing sweeps — the per-level body of the reference's cycle loop,
// src/euler3d_cpu_double.cpp:383-508 — with the coupling a partitioned level needs: one all-reduce(MIN) of the time
// step per sweep (src/Kernels/cfd_loops.cpp:137-150) and one halo message per neighbour after every Runge-Kutta stage.
// Per stage: the tiles next to ghost nodes first, ONE pack launch for all peers, the message on its way (RCCL:
// ncclSend/ncclRecv grouped on a second stream; in-process groups: hipMemcpyPeerAsync), the interior tiles meanwhile,
// then ONE unpack launch before the next stage's boundary tiles.  Same arithmetic as mgcfd_sweep_stage, bit for bit.
// ==========================================================================================================
namespace {

// ---- RCCL, loaded at run time (only a multi-process run needs it; a process that already holds torch's copy gets that one)
struct Rccl {
    struct Id { char b[128]; };                   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;    // (optional: what the communicator itself says about its size and this rank)
    int (*CommUserRank)(void *, int *) = nullptr;
    static constexpr int kDouble = 8, kMin = 3, kSum = 0;     // ncclDouble, ncclMin, ncclSum (rccl.h)
};
Rccl g_rccl;

void rccl_load()
{
    if (g_rccl.lib) return;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) throw std::runtime_error(std::string("cannot load librccl: ") + dlerror());
    auto sym = [&](const char *n) { void *p = dlsym(h, n); if (!p) throw std::runtime_error(std::string("librccl lacks ") + n); return p; };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
    g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(sym("ncclAllReduce"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(dlsym(h, "ncclCommCount"));
    g_rccl.CommUserRank = reinterpret_cast<decltype(g_rccl.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
    g_rccl.lib = h;
}
#define RCCL_CHECK(x) do { const int rc_ = (x); if (rc_ != 0) throw std::runtime_error(std::string("RCCL: ") + #x + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc_) : "error")); } while (0)

} // namespace

// what a solver knows about the ranks around it
struct mgcfd_comm {
    int rank = 0, world = 1;
    void *rccl = nullptr;                         // ncclComm_t (one rank per process)
    struct mgcfd_group *group = nullptr;          // or: the in-process group this solver is rank `rank` of
};
struct mgcfd_group {
    std::vector<mgcfd_solver *> ranks;
    bool peer_ok = true;                          // every device of the group can address every other one's memory (direct mode stores into it)
};

static std::map<mgcfd_solver *, mgcfd_comm> g_comms;     // (one host thread per solver; groups are driven by one thread)

static mgcfd_comm &comm_of(mgcfd_solver *s)
{
    auto it = g_comms.find(s);
    if (it == g_comms.end()) throw std::invalid_argument("the solver is not a rank of anything: call mgcfd_rank_attach_rccl or mgcfd_group_create first");
    return it->second;
}

static void build_halo(mgcfd_solver *s, int level, int n_peers, const int *peers, const int64_t *send_counts, const int64_t *const *send_ids,
                       const int64_t *recv_counts, const int64_t *const *recv_ids, int world)
{
    s->use_device();
    DeviceLevel &lv = s->level(level);
    auto hx = std::make_unique<HaloExchange>();
    std::vector<int32_t> sidx, ridx;
    hx->send_off.push_back(0); hx->recv_off.push_back(0);
    for (int k = 0; k < n_peers; k++) {
        if (k > 0 && peers[k] <= peers[k - 1]) throw std::invalid_argument("peers must be ascending");
        hx->peer.push_back(peers[k]);
        for (int64_t i = 0; i < send_counts[k]; i++) {
            const int64_t id = send_ids[k][i];
            if (id < 0 || id >= lv.n_owned) throw std::invalid_argument("a node sent to a peer must be owned");
            sidx.push_back(lv.plan.new_of_old[static_cast<size_t>(id)]);
        }
        for (int64_t i = 0; i < recv_counts[k]; i++) {
            const int64_t id = recv_ids[k][i];
            if (id < lv.n_owned || id >= lv.info.nel) throw std::invalid_argument("a node received from a peer must be a ghost");
            ridx.push_back(lv.plan.new_of_old[static_cast<size_t>(id)]);
        }
        hx->send_off.push_back(static_cast<int64_t>(sidx.size()));
        hx->recv_off.push_back(static_cast<int64_t>(ridx.size()));
    }
    hx->send_idx = dev_upload(sidx);
    hx->recv_idx = dev_upload(ridx);
    hx->recv_idx_host = ridx;
    for (int j = 0; j < 3; j++) HIP_CHECK(hipEventCreateWithFlags(&hx->bdone[j], hipEventDisableTiming));
    for (int b = 0; b < HaloExchange::kSets; b++) {
        hx->send_buf[b] = dev_alloc<double>(sidx.size() * 5);
        hx->recv_buf[b] = dev_alloc<double>(ridx.size() * 5);
        HIP_CHECK(hipEventCreateWithFlags(&hx->packed[b], hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&hx->arrived[b], hipEventDisableTiming));
    }
    HIP_CHECK(hipEventCreateWithFlags(&hx->reduced, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&hx->gathered, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&hx->joined, hipEventDisableTiming));
    HIP_CHECK(hipStreamCreateWithFlags(&hx->comm_stream, hipStreamNonBlocking));
    hx->gmin = dev_alloc<double>(static_cast<size_t>(std::max(world, 1)));
    // tiles next to ghosts: a tile that holds a ghost node, or an owned node with an edge to one
    std::vector<char> boundary(static_cast<size_t>(lv.plan.n_tiles), 0);
    auto tile_of = [&](int64_t original) { return lv.plan.new_of_old[static_cast<size_t>(original)] / kTile; };
    for (int64_t g = lv.n_owned; g < lv.info.nel; g++) boundary[static_cast<size_t>(tile_of(g))] = 1;
    for (int64_t e = lv.info.internal_start; e < lv.info.internal_start + lv.info.n_internal; e++) {
        const mgcfd_edge &E = lv.edges[static_cast<size_t>(e)];
        if (E.a >= lv.n_owned || E.b >= lv.n_owned) { boundary[static_cast<size_t>(tile_of(E.a))] = 1; boundary[static_cast<size_t>(tile_of(E.b))] = 1; }
    }
    std::vector<int32_t> tb, ti;
    for (int32_t t = 0; t < lv.plan.n_tiles; t++) {
        if (lv.plan.ghosts_last && int64_t(t) * kTile >= lv.n_owned) break;      // ghost-only tiles are never launched
        (boundary[static_cast<size_t>(t)] ? tb : ti).push_back(t);
    }
    hx->n_boundary = static_cast<int32_t>(tb.size());
    hx->n_interior = static_cast<int32_t>(ti.size());
    hx->tiles_boundary = dev_upload(tb);
    hx->tiles_interior = dev_upload(ti);
    {
        std::vector<int32_t> all(tb);
        all.insert(all.end(), ti.begin(), ti.end());
        hx->tiles_all = dev_upload(all);
    }
    lv.hx = std::move(hx);
}

// the message of the state `field` (the buffer a stage just wrote, or `variables`) in buffer set b: pack, start the transfer
static void halo_start(mgcfd_solver *s, int level, const double *field, int b)
{
    DeviceLevel &lv = s->level(level);
    HaloExchange &hx = *lv.hx;
    mgcfd_comm &c = comm_of(s);
    if (hx.total_send() > 0)
        exact::launch_halo_pack(s->stream, hx.total_send(), lv.dp.stride, hx.send_idx, field, hx.send_buf[b]);
    HIP_CHECK(hipEventRecord(hx.packed[b], s->stream));
    if (c.rccl) {
        HIP_CHECK(hipStreamWaitEvent(hx.comm_stream, hx.packed[b], 0));
        RCCL_CHECK(g_rccl.GroupStart());
        for (size_t k = 0; k < hx.peer.size(); k++) {
            const int64_t ns = hx.send_off[k + 1] - hx.send_off[k], nr = hx.recv_off[k + 1] - hx.recv_off[k];
            if (ns > 0) RCCL_CHECK(g_rccl.Send(hx.send_buf[b] + hx.send_off[k] * 5, static_cast<size_t>(ns) * 5, Rccl::kDouble, hx.peer[k], c.rccl, hx.comm_stream));
            if (nr > 0) RCCL_CHECK(g_rccl.Recv(hx.recv_buf[b] + hx.recv_off[k] * 5, static_cast<size_t>(nr) * 5, Rccl::kDouble, hx.peer[k], c.rccl, hx.comm_stream));
        }
        RCCL_CHECK(g_rccl.GroupEnd());
        HIP_CHECK(hipEventRecord(hx.arrived[b], hx.comm_stream));
    }
    // (in-process group: the copies are issued by group_deliver once every rank has packed)
}

// in-process group: every rank's segments copied into its peers' receive buffers (device to device, over xGMI between
// devices), each destination's copies on its own comm stream behind the sources' pack events
static void deliver_to(mgcfd_group *g, mgcfd_solver *dst, int level, int b, bool behind_own_pack);
static void group_deliver(mgcfd_group *g, int level, int b, bool behind_own_pack = false)
{
    for (mgcfd_solver *dst : g->ranks) deliver_to(g, dst, level, b, behind_own_pack);
}
// the segments the other ranks packed for `dst` into its receive buffer (its comm stream behind the sources' pack events)
static void deliver_to(mgcfd_group *g, mgcfd_solver *dst, int level, int b, bool behind_own_pack)
{
    {
        dst->use_device();
        DeviceLevel &ld = dst->level(level);
        HaloExchange &hd = *ld.hx;
        const int me = comm_of(dst).rank;
        bool any = false;
        // (an exchange outside the sweeps' three-set rotation: the destination's receive buffer is free only behind its own
        //  stream's last unpack, which lies before its pack of this exchange)
        if (behind_own_pack) HIP_CHECK(hipStreamWaitEvent(hd.comm_stream, hd.packed[b], 0));
        for (size_t k = 0; k < hd.peer.size(); k++) {
            mgcfd_solver *src = g->ranks[static_cast<size_t>(hd.peer[k])];
            HaloExchange &hs = *src->level(level).hx;
            const auto it = std::find(hs.peer.begin(), hs.peer.end(), me);
            if (it == hs.peer.end()) throw std::logic_error("halo lists of two ranks do not match");
            const size_t ks = static_cast<size_t>(it - hs.peer.begin());
            const int64_t n = hd.recv_off[k + 1] - hd.recv_off[k];
            if (n != hs.send_off[ks + 1] - hs.send_off[ks]) throw std::logic_error("halo message lengths of two ranks do not match");
            if (n == 0) continue;
            HIP_CHECK(hipStreamWaitEvent(hd.comm_stream, hs.packed[b], 0));
            // (unified addressing + peer access: a plain device-to-device copy crosses xGMI; unlike hipMemcpyPeerAsync it can be captured)
            HIP_CHECK(hipMemcpyAsync(hd.recv_buf[b] + hd.recv_off[k] * 5, hs.send_buf[b] + hs.send_off[ks] * 5,
                                     sizeof(double) * 5 * static_cast<size_t>(n), hipMemcpyDeviceToDevice, hd.comm_stream));
            any = true;
        }
        if (!any) HIP_CHECK(hipStreamWaitEvent(hd.comm_stream, hd.packed[b], 0));     // (keeps the comm stream behind the rank's own stream)
        HIP_CHECK(hipEventRecord(hd.arrived[b], hd.comm_stream));
    }
}

// wait for the message of buffer set b and write it into the ghosts of `field`
static void halo_finish(mgcfd_solver *s, int level, double *field, int b)
{
    DeviceLevel &lv = s->level(level);
    HaloExchange &hx = *lv.hx;
    HIP_CHECK(hipStreamWaitEvent(s->stream, hx.arrived[b], 0));
    if (hx.total_recv() > 0)
        exact::launch_halo_unpack(s->stream, hx.total_recv(), lv.dp.stride, hx.recv_idx, hx.recv_buf[b], field);
}

// (MGCFD_PART_LOOK_AHEAD=0: the last stage of a partitioned sweep does not compute the next sweep's step-factor minima —
//  every sweep then starts with its own k_step_factor_local launch; for A/B measurements)
static bool part_look_ahead()
{
    static const bool on = !(std::getenv("MGCFD_PART_LOOK_AHEAD") && std::atoi(std::getenv("MGCFD_PART_LOOK_AHEAD")) == 0);
    return on;
}

// One stage of a partitioned sweep on one rank, part 1: the ghosts of the stage's input arrive (the previous stage's
// message), the boundary tiles run, their results are packed and sent.  Part 2 (stage_interior) runs the other tiles
// while the message travels.
static void stage_boundary(mgcfd_solver *s, int level, int j, int apply_min, const double *min_list, int n_min)
{
    DeviceLevel &lv = s->level(level);
    HaloExchange &hx = *lv.hx;
    double *const start = lv.q, *const b1 = lv.q_alt, *const b2 = lv.old_variables;
    double *in = j == 0 ? start : (j == 1 ? b1 : b2);
    double *out = j == 0 ? b1 : (j == 1 ? b2 : b1);
    if (j > 0) halo_finish(s, level, in, j - 1);
    if (j == 0) {
        if (!lv.fluxes_zero) throw std::invalid_argument("a partitioned sweep needs zero fluxes (as after time_step)");
        s->settle_fluxes(lv);
    }
    s->force_check = s->next_check();                       // both parts of the stage are one time_step for check_for_invalid_variables
    s->op_fused_stage(level, j, in, out, j == 0 ? apply_min : 0, j == 2, start, j == 2 && apply_min != 0 && part_look_ahead(), false, nullptr, hx.tiles_boundary, hx.n_boundary, true, min_list, n_min);
    lv.stage_out = out;
    halo_start(s, level, out, j);
}
static void stage_interior(mgcfd_solver *s, int level, int j, int apply_min, const double *min_list, int n_min)
{
    DeviceLevel &lv = s->level(level);
    HaloExchange &hx = *lv.hx;
    double *const start = lv.q, *const b1 = lv.q_alt, *const b2 = lv.old_variables;
    double *in = j == 0 ? start : (j == 1 ? b1 : b2);
    double *out = j == 0 ? b1 : (j == 1 ? b2 : b1);
    s->op_fused_stage(level, j, in, out, j == 0 ? apply_min : 0, j == 2, start, j == 2 && apply_min != 0 && part_look_ahead(), false, nullptr, hx.tiles_interior, hx.n_interior, false, min_list, n_min);
    s->force_check = -1;
    if (j == 2) {
        lv.rot = (lv.rot + 1) % 3;                          // variables = b1, q_alt = b2, old_variables = start
        lv.apply_rot();
        lv.stage_out = lv.q;
        // global time step: both launches of the last stage left the first half of the NEXT sweep's compute_step_factor
        // in partial_min, every tile its own minimum over its OWNED nodes (cbrt_vol is +inf on ghosts)
        lv.min_ahead = apply_min != 0 && part_look_ahead();
    }
}

static void sweep_first_half(mgcfd_solver *s, int level, double *min_out = nullptr)
{
    DeviceLevel &lv = s->level(level);
    if (!lv.hx) throw std::invalid_argument("the level has no halo lists: call mgcfd_rank_set_halo first");
    if (lv.stage_next != 0) throw std::invalid_argument("a sweep is under way (mgcfd_sweep_stage)");
    const bool global_dt = s->mesh_variant != MGCFD_MESH_FVCORR;
    if (global_dt && lv.min_ahead) lv.iters[MGCFD_LOOP_COMPUTE_STEP] += lv.info.nel;      // the previous sweep's last stage looked ahead
    else s->op_step_factor(level, true, false);             // first half of compute_step_factor (a ghost's factor is its owner's business)
    if (global_dt) exact::launch_min_reduce(s->stream, lv.info.nel, lv.partial_min, min_out ? min_out : lv.min_dt);
}

// one sweep of an RCCL rank, as issued (and as captured): every exchange it starts it also finishes
static void rank_sweep_once(mgcfd_solver *s, int level)
{
    mgcfd_comm &c = comm_of(s);
    DeviceLevel &lv = s->level(level);
    const bool global_dt = s->mesh_variant != MGCFD_MESH_FVCORR;
    sweep_first_half(s, level);
    if (global_dt) RCCL_CHECK(g_rccl.AllReduce(lv.min_dt, lv.min_dt, 1, Rccl::kDouble, Rccl::kMin, c.rccl, s->stream));
    for (int j = 0; j < MGCFD_RK; j++) {
        stage_boundary(s, level, j, global_dt ? 2 : 0, nullptr, 0);
        stage_interior(s, level, j, global_dt ? 2 : 0, nullptr, 0);
    }
    halo_finish(s, level, lv.q, MGCFD_RK - 1);              // the ghosts of `variables` are current when the sweep ends
}

// ---- in-process groups, direct mode -------------------------------------------------------------------------------------
// Every rank of a group can store into every other rank's memory (one process: unified addressing, peer access over xGMI),
// so a message needs no buffers: behind its boundary tiles a rank runs ONE k_halo_push that writes the nodes its peers need
// straight into their ghost slots of the buffer the stage writes, and records bdone[stage].  A peer's next stage waits
// for that event before its boundary tiles (the only ones that read ghosts) start.  Ghost slots are written by pushes
// only (the stage launches are told nel = n_owned: the plan numbers ghosts last), so a push may land while the
// receiver's own kernels of that stage still run.  Why nothing is overwritten while somebody may still read it: rank A's
// push of stage j goes into the buffer B's stage j writes, whose ghost slots B last read in the stage before (as that
// stage's input) — and A's boundary tiles of stage j start only behind B's bdone of that earlier stage.
// Per stage and rank: [P event waits] boundary launch, push launch, event record, interior launch — against
// unpack, boundary, pack, record, wait, P copies, record, interior, wait of the buffered form.
// (MGCFD_GROUP_DIRECT=0 keeps the buffered form.)
static bool group_direct_wanted()
{
    // (not together with group graphs, MGCFD_GROUP_GRAPH=1: the capture code joins the buffered form's streams)
    const bool on = !(std::getenv("MGCFD_GROUP_DIRECT") && std::atoi(std::getenv("MGCFD_GROUP_DIRECT")) == 0) &&
                           !(std::getenv("MGCFD_GROUP_GRAPH") && std::atoi(std::getenv("MGCFD_GROUP_GRAPH")) != 0);
    return on;
}

// build the push targets of every rank (once; allocations and uploads never happen inside a sweep)
static void group_prepare_direct(mgcfd_group *g, int level)
{
    bool possible = group_direct_wanted() && g->peer_ok;   // (a kernel must never store into memory its device cannot address)
    for (mgcfd_solver *s : g->ranks) {
        DeviceLevel &lv = s->level(level);
        if (!lv.hx) throw std::invalid_argument("a rank has no halo lists");
        if (lv.hx->direct) return;                          // (done before)
        if (static_cast<int>(lv.hx->peer.size()) > kMaxPushPeers) possible = false;
        if (lv.n_owned < lv.info.nel && !lv.plan.ghosts_last) possible = false;     // (a plan that mixes ghosts into the tiles: its stages write them)
    }
    if (!possible) return;
    for (mgcfd_solver *src : g->ranks) {
        src->use_device();
        HaloExchange &hs = *src->level(level).hx;
        const int me = comm_of(src).rank;
        std::vector<int32_t> target(static_cast<size_t>(hs.total_send()), 0);
        for (size_t k = 0; k < hs.peer.size(); k++) {
            mgcfd_solver *dst = g->ranks[static_cast<size_t>(hs.peer[k])];
            HaloExchange &hd = *dst->level(level).hx;
            const auto it = std::find(hd.peer.begin(), hd.peer.end(), me);
            if (it == hd.peer.end()) throw std::logic_error("halo lists of two ranks do not match");
            const size_t kd = static_cast<size_t>(it - hd.peer.begin());
            const int64_t n = hs.send_off[k + 1] - hs.send_off[k];
            if (n != hd.recv_off[kd + 1] - hd.recv_off[kd]) throw std::logic_error("halo message lengths of two ranks do not match");
            for (int64_t i = 0; i < n; i++)
                target[static_cast<size_t>(hs.send_off[k] + i)] = hd.recv_idx_host[static_cast<size_t>(hd.recv_off[kd] + i)];
        }
        hs.push_target = dev_upload(target);
    }
    for (mgcfd_solver *s : g->ranks) s->level(level).hx->direct = true;
}

// which of a rank's three state buffers a stage writes (as stage_boundary / stage_interior name them)
static double *stage_out_buffer(DeviceLevel &lv, int j) { return j == 1 ? lv.old_variables : lv.q_alt; }

// where a rank's push goes: the buffer `peer_field(peer's level)` of every peer, the peers' segments of the message
template <typename PeerField>
static PushPeers make_push_peers(mgcfd_group *g, mgcfd_solver *s, int level, PeerField &&peer_field)
{
    HaloExchange &hx = *s->level(level).hx;
    PushPeers pp;
    pp.n = static_cast<int>(hx.peer.size());
    for (int k = 0; k < pp.n; k++) {
        DeviceLevel &pl = g->ranks[static_cast<size_t>(hx.peer[static_cast<size_t>(k)])]->level(level);
        pp.base[k] = peer_field(pl);
        pp.stride[k] = pl.dp.stride;
        pp.first[k] = hx.send_off[static_cast<size_t>(k)];
    }
    pp.first[pp.n] = hx.total_send();
    return pp;
}

// the nodes the peers need of `field` (this rank's buffer) into the peers' buffers, then the event
static void push_and_record(mgcfd_solver *s, int level, const double *field, const PushPeers &pp, int ev)
{
    DeviceLevel &lv = s->level(level);
    HaloExchange &hx = *lv.hx;
    if (hx.total_send() > 0) exact::launch_halo_push(s->stream, hx.total_send(), lv.dp.stride, hx.send_idx, hx.push_target, field, pp);
    HIP_CHECK(hipEventRecord(hx.bdone[ev], s->stream));
}

static void wait_for_peers(mgcfd_group *g, mgcfd_solver *s, int level, int ev)
{
    HaloExchange &hx = *s->level(level).hx;
    for (int p : hx.peer) HIP_CHECK(hipStreamWaitEvent(s->stream, g->ranks[static_cast<size_t>(p)]->level(level).hx->bdone[ev], 0));
}

// part 1 of a stage in direct mode (part 2 is stage_interior as it is); `to` = where the push goes (nullptr: the peers'
// buffers as they are named now — one host thread drives the group, nobody has rotated since the stage began)
static void stage_boundary_direct(mgcfd_group *g, mgcfd_solver *s, int level, int j, int apply_min, const double *min_list, int n_min,
                                  const PushPeers *to = nullptr)
{
    DeviceLevel &lv = s->level(level);
    HaloExchange &hx = *lv.hx;
    double *const start = lv.q, *const b1 = lv.q_alt, *const b2 = lv.old_variables;
    double *in = j == 0 ? start : (j == 1 ? b1 : b2);
    double *out = j == 0 ? b1 : (j == 1 ? b2 : b1);
    wait_for_peers(g, s, level, (j + 2) % 3);               // the ghosts of `in`: the peers' pushes of the stage before (stage 0: of the last sweep, or of the exchange)
    if (j == 0) {
        if (!lv.fluxes_zero) throw std::invalid_argument("a partitioned sweep needs zero fluxes (as after time_step)");
        s->settle_fluxes(lv);
    }
    s->force_check = s->next_check();
    s->op_fused_stage(level, j, in, out, j == 0 ? apply_min : 0, j == 2, start, j == 2 && apply_min != 0 && part_look_ahead(), false, nullptr, hx.tiles_boundary, hx.n_boundary, true, min_list, n_min);
    lv.stage_out = out;
    push_and_record(s, level, out, to ? *to : make_push_peers(g, s, level, [&](DeviceLevel &pl) { return stage_out_buffer(pl, j); }), j);
}

// ---- a host thread per rank ---------------------------------------------------------------------------------------------
// One thread issuing every rank's calls costs the host N x ~67 us per sweep (launches 3.5 us, event records and waits 4-5 us
// each on ROCm) — more than the kernels of a rank take.  In direct mode every rank's calls touch only its own stream, its
// own events and its own solver, so each rank gets a thread of its own.  What the threads must agree on is the ORDER of
// event records and waits (a wait refers to the event's most recent record at the time of the call): a barrier after the
// first half of compute_step_factor and after every stage puts each record ahead of the waits for it.
namespace {
struct SpinBarrier {
    explicit SpinBarrier(int n) : n_(n) {}
    void wait()
    {
        const unsigned gen = gen_.load(std::memory_order_acquire);
        if (count_.fetch_add(1, std::memory_order_acq_rel) + 1 == n_) {
            count_.store(0, std::memory_order_relaxed);
            gen_.fetch_add(1, std::memory_order_release);
        } else {
            for (int spins = 0; gen_.load(std::memory_order_acquire) == gen; spins++)
                if (spins > 4000) std::this_thread::yield();
        }
    }
    std::atomic<int> count_{0};
    std::atomic<unsigned> gen_{0};
    const int n_;
};
}

// `sweeps` sweeps of every rank, a thread per rank (the calling thread takes rank 0).  with_rms: after every sweep each rank
// appends the sum of its owned nodes' squared residuals to its ring (mgcfd_solver::rms_ring) for one read-back at the end.
static void group_sweeps_threaded(mgcfd_group *g, int level, int sweeps, bool with_rms)
{
    const int n = static_cast<int>(g->ranks.size());
    SpinBarrier bar(n);
    std::atomic<bool> failed{false};
    std::mutex mu;
    std::exception_ptr first_error;
    const bool global_dt = g->ranks[0]->mesh_variant != MGCFD_MESH_FVCORR;
    auto run = [&](int r) {
        mgcfd_solver *s = g->ranks[static_cast<size_t>(r)];
        // (a rank that failed keeps arriving at the barriers, doing nothing, so that the others are not left waiting)
        auto step = [&](auto &&body) {
            if (failed.load(std::memory_order_acquire)) return;
            try { body(); }
            catch (...) { std::lock_guard<std::mutex> lock(mu); if (!first_error) first_error = std::current_exception(); failed.store(true, std::memory_order_release); }
        };
        step([&] { s->use_device(); });
        DeviceLevel &lv = s->level(level);
        HaloExchange &hx = *lv.hx;
        for (int k = 0; k < sweeps; k++) {
            const int par = hx.min_parity;
            step([&] {
                sweep_first_half(s, level, hx.min_par + par);
                if (global_dt) HIP_CHECK(hipEventRecord(hx.reduced, s->stream));
            });
            hx.min_parity = par ^ 1;
            bar.wait();
            PushPeers to[MGCFD_RK];
            step([&] {
                if (global_dt) {
                    for (mgcfd_solver *src : g->ranks) if (src != s) HIP_CHECK(hipStreamWaitEvent(s->stream, src->level(level).hx->reduced, 0));
                    exact::launch_min_over_peers(s->stream, hx.peer_scalars[par], n, hx.gmin);
                }
                // the peers' buffers of this sweep's three stages, read while nobody rotates (a rank rotates in its last stage's
                // second part, two barriers from here; its previous rotation lies before the barrier just passed)
                for (int j = 0; j < MGCFD_RK; j++) to[j] = make_push_peers(g, s, level, [&](DeviceLevel &pl) { return stage_out_buffer(pl, j); });
            });
            for (int j = 0; j < MGCFD_RK; j++) {
                step([&] {
                    stage_boundary_direct(g, s, level, j, global_dt ? 3 : 0, hx.gmin, 1, &to[j]);
                    stage_interior(s, level, j, global_dt ? 3 : 0, hx.gmin, 1);
                });
                bar.wait();
            }
            if (with_rms) step([&] {
                exact::launch_sumsq(s->stream, lv.info.nel, lv.dp.stride, lv.residuals, lv.partials, lv.n_partials, lv.sumsq, lv.dp.old_of_new, lv.n_owned);
                exact::launch_append_scalar(s->stream, lv.sumsq, s->rms_ring, s->rms_count, mgcfd_solver::kRmsRing);
            });
        }
        step([&] { wait_for_peers(g, s, level, 2); HIP_CHECK(hipGetLastError()); });
    };
    std::vector<std::thread> threads;
    for (int r = 1; r < n; r++) threads.emplace_back(run, r);
    run(0);
    for (std::thread &t : threads) t.join();
    if (first_error) std::rethrow_exception(first_error);
}


// one sweep of every rank of an in-process group
static void group_sweep_once(mgcfd_group *g, int level)
{
    const int n = static_cast<int>(g->ranks.size());
    const bool global_dt = g->ranks[0]->mesh_variant != MGCFD_MESH_FVCORR;
    const int par = g->ranks[0]->level(level).hx->min_parity;       // (the ranks of a group sweep in step)
    for (mgcfd_solver *s : g->ranks) { s->use_device(); HaloExchange &hx = *s->level(level).hx; sweep_first_half(s, level, hx.min_par + par); hx.min_parity = par ^ 1; }
    if (global_dt) {
        // all-reduce(MIN) of one fp64 over the group: behind every rank's reduction event each rank reads the others'
        // scalars itself (k_min_over_peers: 8-byte loads over xGMI) — no message, no second stream
        for (mgcfd_solver *src : g->ranks) { src->use_device(); HIP_CHECK(hipEventRecord(src->level(level).hx->reduced, src->stream)); }
        for (mgcfd_solver *dst : g->ranks) {
            dst->use_device();
            HaloExchange &hd = *dst->level(level).hx;
            for (mgcfd_solver *src : g->ranks) if (src != dst) HIP_CHECK(hipStreamWaitEvent(dst->stream, src->level(level).hx->reduced, 0));
            exact::launch_min_over_peers(dst->stream, hd.peer_scalars[par], n, hd.gmin);
        }
    }
    const bool direct = g->ranks[0]->level(level).hx->direct;
    for (int j = 0; j < MGCFD_RK; j++) {
        for (mgcfd_solver *s : g->ranks) {
            s->use_device();
            if (direct) stage_boundary_direct(g, s, level, j, global_dt ? 3 : 0, s->level(level).hx->gmin, 1);
            else stage_boundary(s, level, j, global_dt ? 3 : 0, s->level(level).hx->gmin, 1);
        }
        // (set 0 is also the set of the transfers' exchanges, outside the sweeps' rotation: a destination's copies of stage 0 wait
        //  for its OWN pack of this stage, which lies behind its unpack of such an exchange — with the local time step nothing
        //  else orders a peer's stage-0 message behind that unpack; round 3's advisor finding)
        if (!direct) group_deliver(g, level, j, j == 0);
        for (mgcfd_solver *s : g->ranks) { s->use_device(); stage_interior(s, level, j, global_dt ? 3 : 0, s->level(level).hx->gmin, 1); }
    }
    // (direct mode: the last stage's pushes went into the buffer that is `variables` now; whoever reads ghosts next waits for bdone[2])
    if (!direct) for (mgcfd_solver *s : g->ranks) { s->use_device(); halo_finish(s, level, s->level(level).q, MGCFD_RK - 1); }
}

// The sweep as a hipGraph: the ~25 host calls a rank's sweep takes (launches, event records and waits, copies or RCCL
// calls) cost more host time than the kernels take on the GPU; captured once per buffer rotation (the launch arguments
// repeat with period 3) a sweep is ONE host call.  `body` issues the sweep; `streams` beyond the first are forked
// from and joined to the capturing stream.  Returns false (and leaves everything eager) when capture is not possible.
template <typename Body>
static bool capture_sweep(hipStream_t origin, const std::vector<std::pair<hipStream_t, hipEvent_t>> &others, hipEvent_t fork, Body &&body,
                          hipGraphExec_t *exec)
{
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(origin, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); return false; }
    bool ok = true;
    try {
        HIP_CHECK(hipEventRecord(fork, origin));
        for (auto &o : others) HIP_CHECK(hipStreamWaitEvent(o.first, fork, 0));
        body();
        for (auto &o : others) { HIP_CHECK(hipEventRecord(o.second, o.first)); HIP_CHECK(hipStreamWaitEvent(origin, o.second, 0)); }
    } catch (...) {
        ok = false;
    }
    if (hipStreamEndCapture(origin, &graph) != hipSuccess || !graph) { (void)hipGetLastError(); ok = false; }
    if (ok && hipGraphInstantiate(exec, graph, nullptr, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); *exec = nullptr; ok = false; }
    if (graph) (void)hipGraphDestroy(graph);
    return ok;
}


// ---- ranks in different processes, direct mode (HIP IPC) -------------------------------------------------------------------
// What the in-process groups do with events, done between processes: every rank publishes IPC handles of its three state
// buffers and of a small array of flag words (mgcfd_rank_ipc_export), opens its neighbours' (mgcfd_rank_ipc_attach), and from
// then on a stage is: k_flags_wait (the neighbours' messages of the stage before have arrived) -> boundary tiles -> ONE
// k_halo_push_flags (stores into the neighbours' ghost slots, then raises their flags) -> interior tiles.  No message
// buffers, no second stream, no RCCL call per stage; the all-reduce of the global time step stays on RCCL.
// Opt-in (bench.py --exchange ipc): it could be rehearsed with two processes on ONE GPU only (tests/test_gpu_configs.py).
namespace {
struct IpcExportHeader {
    hipIpcMemHandle_t state[3], flags, gmins;
    int64_t stride, n_recv;
    int32_t rot, rank, n_peers, reserved;
    int32_t peers[kMaxPushPeers];
    int64_t recv_off[kMaxPushPeers + 1];
};
}

static void ipc_wait(mgcfd_solver *s, HaloExchange &hx, int slot)
{
    FlagRows rows;
    rows.n = static_cast<int>(hx.peer.size());
    for (int k = 0; k < rows.n; k++) rows.row[k] = hx.peer[static_cast<size_t>(k)];       // (a rank's row is its rank)
    exact::launch_flags_wait(s->stream, hx.flags, rows, slot, hx.seq, hx.ipc_timeouts);
}

// all-reduce(MIN) of lv.min_dt over every rank through the flags: the minima end up in hx.gmins[parity][0..world)
static const double *ipc_allreduce_min(mgcfd_solver *s, DeviceLevel &lv)
{
    HaloExchange &hx = *lv.hx;
    const mgcfd_comm &c = comm_of(s);
    MinPublish mp;
    mp.world = c.world; mp.me = c.rank;
    mp.parity = static_cast<int>(hx.sweep_seq & 1ull);
    hx.sweep_seq++;
    mp.value = hx.sweep_seq;
    FlagRows rows;
    for (int r = 0; r < c.world; r++) {
        mp.mins[r] = hx.all_mins[r];
        mp.flag[r] = hx.all_flag[r];
        if (r != c.rank) rows.row[rows.n++] = r;
    }
    exact::launch_min_publish(s->stream, lv.min_dt, mp);
    exact::launch_flags_wait(s->stream, hx.flags, rows, 3, hx.sweep_seq, hx.ipc_timeouts);
    return hx.gmins + mp.parity * kMaxIpcRanks;
}

// this rank's nodes of `field` into the peers' buffers state[(rot + which) % 3] (which: 0 variables, 1 q_alt, 2 old_variables,
// by the PEER's numbering of its buffers), flag word `slot`
static void ipc_push(mgcfd_solver *s, DeviceLevel &lv, const double *field, int which, int slot)
{
    HaloExchange &hx = *lv.hx;
    PushPeers pp;
    PushFlags pf;
    pp.n = pf.n = static_cast<int>(hx.peer.size());
    for (int k = 0; k < pp.n; k++) {
        pp.base[k] = hx.peer_state[k][(lv.rot + hx.peer_rot_delta[k] + which) % 3];
        pp.stride[k] = hx.peer_stride[k];
        pp.first[k] = hx.send_off[static_cast<size_t>(k)];
        pf.flag[k] = hx.peer_flag[k] + slot;
    }
    pp.first[pp.n] = hx.total_send();
    hx.seq++;
    pf.value = hx.seq;
    exact::launch_halo_push_flags(s->stream, hx.total_send(), lv.dp.stride, hx.send_idx, hx.push_target, field, pp, pf, hx.ticket);
}

static void rank_sweep_once_ipc(mgcfd_solver *s, int level)
{
    mgcfd_comm &c = comm_of(s);
    DeviceLevel &lv = s->level(level);
    HaloExchange &hx = *lv.hx;
    const bool global_dt = s->mesh_variant != MGCFD_MESH_FVCORR;
    if (global_dt && !c.rccl && !hx.ipc_all) throw std::invalid_argument("a global time step needs the all-reduce: attach every rank (mgcfd_rank_ipc_attach with all exports) or RCCL");
    sweep_first_half(s, level);
    // the time step: over the flags when every rank is attached (the ranks' minima side by side, the first stage takes their
    // minimum), else RCCL's all-reduce into the scalar
    const double *min_list = nullptr;
    int n_min = 0, apply = 0;
    if (global_dt) {
        if (hx.ipc_all) { min_list = ipc_allreduce_min(s, lv); n_min = c.world; apply = 3; }
        else { RCCL_CHECK(g_rccl.AllReduce(lv.min_dt, lv.min_dt, 1, Rccl::kDouble, Rccl::kMin, c.rccl, s->stream)); apply = 2; }
    }
    for (int j = 0; j < MGCFD_RK; j++) {
        double *const start = lv.q, *const b1 = lv.q_alt, *const b2 = lv.old_variables;
        double *in = j == 0 ? start : (j == 1 ? b1 : b2);
        double *out = j == 0 ? b1 : (j == 1 ? b2 : b1);
        ipc_wait(s, hx, (j + 2) % 3);                       // the ghosts of `in`: the peers' pushes of the stage before (stage 0: of the last sweep, or of the exchange)
        if (j == 0) {
            if (!lv.fluxes_zero) throw std::invalid_argument("a partitioned sweep needs zero fluxes (as after time_step)");
            s->settle_fluxes(lv);
        }
        s->force_check = s->next_check();
        if (s->opt_rank_split == 2 && hx.n_boundary > 0) {
            // MGCFD_OPT_RANK_SPLIT = 2: ONE launch per stage that sends its own message — the boundary tiles first, their
            // epilogue stores into the neighbours, the last of them raises the flags, the interior tiles run on meanwhile
            StagePush sp;
            sp.send_ptr = hx.node_send_ptr; sp.send_peer = hx.node_send_peer; sp.send_target = hx.node_send_target;
            sp.ticket = hx.ticket;
            sp.n_boundary = hx.n_boundary;
            const int which = j == 1 ? 2 : 1;
            sp.peers.n = sp.flags.n = static_cast<int>(hx.peer.size());
            for (int k = 0; k < sp.peers.n; k++) {
                sp.peers.base[k] = hx.peer_state[k][(lv.rot + hx.peer_rot_delta[k] + which) % 3];
                sp.peers.stride[k] = hx.peer_stride[k];
                sp.flags.flag[k] = hx.peer_flag[k] + j;
            }
            hx.seq++;
            sp.flags.value = hx.seq;
            s->op_fused_stage(level, j, in, out, j == 0 ? apply : 0, j == 2, start, j == 2 && global_dt && part_look_ahead(), false, nullptr, hx.tiles_all, hx.n_boundary + hx.n_interior, true, min_list, n_min, false, &sp);
            s->force_check = -1;
            lv.stage_out = out;
            if (j == 2) {
                lv.rot = (lv.rot + 1) % 3;
                lv.apply_rot();
                lv.stage_out = lv.q;
                lv.min_ahead = apply != 0 && part_look_ahead();
            }
        } else if (s->opt_rank_split) {
            s->op_fused_stage(level, j, in, out, j == 0 ? apply : 0, j == 2, start, j == 2 && global_dt && part_look_ahead(), false, nullptr, hx.tiles_boundary, hx.n_boundary, true, min_list, n_min);
            lv.stage_out = out;
            ipc_push(s, lv, out, j == 1 ? 2 : 1, j);        // (before the rotation: stage 1 writes old_variables' buffer, stages 0 and 2 q_alt's)
            stage_interior(s, level, j, apply, min_list, n_min);
        } else {
            // MGCFD_OPT_RANK_SPLIT = 0: every tile of the rank in one launch, the message behind it
            s->op_fused_stage(level, j, in, out, j == 0 ? apply : 0, j == 2, start, j == 2 && global_dt && part_look_ahead(), false, nullptr, hx.tiles_all, hx.n_boundary + hx.n_interior, true, min_list, n_min);
            s->force_check = -1;
            lv.stage_out = out;
            ipc_push(s, lv, out, j == 1 ? 2 : 1, j);
            if (j == 2) {
                lv.rot = (lv.rot + 1) % 3;
                lv.apply_rot();
                lv.stage_out = lv.q;
                lv.min_ahead = apply != 0 && part_look_ahead();
            }
        }
    }
}

extern "C" {

int mgcfd_rccl_unique_id(void *out128)
{
    REQUIRE(out128);
    return guarded([&] { rccl_load(); RCCL_CHECK(g_rccl.GetUniqueId(out128)); });
}

int mgcfd_rank_attach_rccl(mgcfd_solver *s, int rank, int world, const void *id128)
{
    REQUIRE(s); REQUIRE(id128);
    return guarded([&] {
        if (rank < 0 || rank >= world) throw std::invalid_argument("rank out of range");
        s->use_device();
        rccl_load();
        Rccl::Id id;
        std::memcpy(id.b, id128, sizeof(id.b));
        mgcfd_comm c;
        c.rank = rank; c.world = world;
        RCCL_CHECK(g_rccl.CommInitRank(&c.rccl, world, id, rank));
        g_comms[s] = c;
    });
}

// a rank of `world` processes without an RCCL communicator: enough for a level with a local time step (mesh_name = fvcorr),
// whose sweeps need no all-reduce, once the neighbours' buffers are attached (mgcfd_rank_ipc_attach)
int mgcfd_rank_attach_plain(mgcfd_solver *s, int rank, int world)
{
    REQUIRE(s);
    return guarded([&] {
        if (rank < 0 || rank >= world) throw std::invalid_argument("rank out of range");
        mgcfd_comm c;
        c.rank = rank; c.world = world;
        g_comms[s] = c;
    });
}

int mgcfd_rank_ipc_export_size(mgcfd_solver *s, int level, int64_t *bytes)
{
    REQUIRE(s); REQUIRE(bytes);
    return guarded([&] {
        DeviceLevel &lv = s->level(level);
        if (!lv.hx) throw std::invalid_argument("the level has no halo lists: call mgcfd_rank_set_halo first");
        *bytes = static_cast<int64_t>(sizeof(IpcExportHeader) + sizeof(int32_t) * lv.hx->recv_idx_host.size());
    });
}

int mgcfd_rank_ipc_export(mgcfd_solver *s, int level, void *out)
{
    REQUIRE(s); REQUIRE(out);
    return guarded([&] {
        s->use_device();
        DeviceLevel &lv = s->level(level);
        if (!lv.hx) throw std::invalid_argument("the level has no halo lists: call mgcfd_rank_set_halo first");
        HaloExchange &hx = *lv.hx;
        if (static_cast<int>(hx.peer.size()) > kMaxPushPeers) throw std::invalid_argument("more neighbouring ranks than a push addresses (8)");
        if (lv.n_owned < lv.info.nel && !lv.plan.ghosts_last) throw std::invalid_argument("the level's plan mixes ghosts into the tiles");
        if (!hx.flags) {
            // the words other devices write and this one polls: fine-grained device memory (not cached across the writes of
            // another agent) where the runtime grants it, ordinary device memory otherwise
            auto fine = [](size_t bytes) {
                void *m = nullptr;
                if (hipExtMallocWithFlags(&m, bytes, hipDeviceMallocFinegrained) != hipSuccess || !m) { (void)hipGetLastError(); HIP_CHECK(hipMalloc(&m, bytes)); }
                return m;
            };
            hx.flags = static_cast<unsigned long long *>(fine(sizeof(unsigned long long) * kMaxIpcRanks * 4));
            HIP_CHECK(hipMemset(hx.flags, 0, sizeof(unsigned long long) * kMaxIpcRanks * 4));
            hx.gmins = static_cast<double *>(fine(sizeof(double) * 2 * kMaxIpcRanks));
            const std::vector<double> inf(2 * kMaxIpcRanks, std::numeric_limits<double>::infinity());
            HIP_CHECK(hipMemcpy(hx.gmins, inf.data(), sizeof(double) * inf.size(), hipMemcpyHostToDevice));
            hx.ticket = dev_alloc<unsigned>(1);
            HIP_CHECK(hipMemset(hx.ticket, 0, sizeof(unsigned)));
            hx.ipc_timeouts = dev_alloc<int>(1);
            HIP_CHECK(hipMemset(hx.ipc_timeouts, 0, sizeof(int)));
        }
        IpcExportHeader h{};
        for (int k = 0; k < 3; k++) HIP_CHECK(hipIpcGetMemHandle(&h.state[k], lv.state[k]));
        HIP_CHECK(hipIpcGetMemHandle(&h.flags, hx.flags));
        HIP_CHECK(hipIpcGetMemHandle(&h.gmins, hx.gmins));
        if (comm_of(s).world > kMaxIpcRanks) throw std::invalid_argument("more ranks than flag rows (16)");
        h.stride = lv.dp.stride;
        h.n_recv = static_cast<int64_t>(hx.recv_idx_host.size());
        h.rot = lv.rot % 3;
        h.rank = comm_of(s).rank;
        h.n_peers = static_cast<int32_t>(hx.peer.size());
        for (size_t k = 0; k < hx.peer.size(); k++) h.peers[k] = hx.peer[k];
        for (size_t k = 0; k <= hx.peer.size(); k++) h.recv_off[k] = hx.recv_off[k];
        std::memcpy(out, &h, sizeof(h));
        std::memcpy(static_cast<char *>(out) + sizeof(h), hx.recv_idx_host.data(), sizeof(int32_t) * hx.recv_idx_host.size());
    });
}

// exports: what other ranks exported, in any order — at least every neighbouring rank's (mgcfd_rank_set_halo's peers); with
// EVERY other rank's the all-reduce of a global time step goes through the flags as well and no RCCL call is left in a sweep
int mgcfd_rank_ipc_attach(mgcfd_solver *s, int level, int n_exports, const void *const *exports)
{
    REQUIRE(s);
    if (n_exports > 0) REQUIRE(exports);
    return guarded([&] {
        s->use_device();
        DeviceLevel &lv = s->level(level);
        if (!lv.hx || !lv.hx->flags) throw std::invalid_argument("export this rank's buffers first (mgcfd_rank_ipc_export)");
        HaloExchange &hx = *lv.hx;
        const mgcfd_comm &c = comm_of(s);
        const int me = c.rank;
        std::vector<int32_t> target(static_cast<size_t>(hx.total_send()), 0);
        std::vector<char> peer_seen(hx.peer.size(), 0), rank_seen(static_cast<size_t>(c.world), 0);
        auto open = [&](const hipIpcMemHandle_t &h) {
            void *m = nullptr;
            HIP_CHECK(hipIpcOpenMemHandle(&m, h, hipIpcMemLazyEnablePeerAccess));
            hx.ipc_opened.push_back(m);
            return m;
        };
        for (int e = 0; e < n_exports; e++) {
            IpcExportHeader h;
            std::memcpy(&h, exports[e], sizeof(h));
            if (h.rank == me) continue;
            if (h.rank < 0 || h.rank >= c.world || rank_seen[static_cast<size_t>(h.rank)]) throw std::invalid_argument("an export of an unknown rank, or of one rank twice");
            rank_seen[static_cast<size_t>(h.rank)] = 1;
            unsigned long long *their_flags = static_cast<unsigned long long *>(open(h.flags));
            hx.all_flag[h.rank] = their_flags + me * 4 + 3;
            hx.all_mins[h.rank] = static_cast<double *>(open(h.gmins));
            const auto it = std::find(hx.peer.begin(), hx.peer.end(), h.rank);
            if (it == hx.peer.end()) continue;              // (not a neighbour: only the all-reduce talks to it)
            const size_t k = static_cast<size_t>(it - hx.peer.begin());
            peer_seen[k] = 1;
            int kd = -1;
            for (int q = 0; q < h.n_peers; q++) if (h.peers[q] == me) kd = q;
            if (kd < 0) throw std::logic_error("halo lists of two ranks do not match");
            const int64_t n = hx.send_off[k + 1] - hx.send_off[k];
            if (n != h.recv_off[kd + 1] - h.recv_off[kd]) throw std::logic_error("halo message lengths of two ranks do not match");
            const int32_t *ridx = reinterpret_cast<const int32_t *>(static_cast<const char *>(exports[e]) + sizeof(h));
            for (int64_t i = 0; i < n; i++) target[static_cast<size_t>(hx.send_off[k] + i)] = ridx[h.recv_off[kd] + i];
            for (int b = 0; b < 3; b++) hx.peer_state[k][b] = static_cast<double *>(open(h.state[b]));
            hx.peer_flag[k] = their_flags + me * 4;
            hx.peer_stride[k] = h.stride;
            hx.peer_rot_delta[k] = ((h.rot - lv.rot % 3) % 3 + 3) % 3;
        }
        for (char seen : peer_seen) if (!seen) throw std::invalid_argument("the export of a neighbouring rank is missing");
        hx.all_mins[me] = hx.gmins;
        hx.all_flag[me] = hx.flags + me * 4 + 3;            // (never raised: a rank does not wait for itself)
        int others = 0;
        for (char seen : rank_seen) others += seen;
        hx.ipc_all = others == c.world - 1;
        if (hx.push_target) { HIP_CHECK(hipFree(hx.push_target)); hx.push_target = nullptr; }
        hx.push_target = dev_upload(target);
        {   // the same message per node (library numbering): what a stage that sends its own message walks (StagePush)
            std::vector<int32_t> sidx(static_cast<size_t>(hx.total_send()));
            HIP_CHECK(hipMemcpy(sidx.data(), hx.send_idx, sizeof(int32_t) * sidx.size(), hipMemcpyDeviceToHost));
            std::vector<int32_t> ptr(static_cast<size_t>(lv.dp.stride) + 2, 0), tgt(sidx.size());
            std::vector<int8_t> peer(sidx.size());
            for (int32_t n : sidx) ptr[static_cast<size_t>(n) + 1]++;
            for (size_t n = 1; n < ptr.size(); n++) ptr[n] += ptr[n - 1];
            std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
            for (size_t k = 0; k < hx.peer.size(); k++)
                for (int64_t m = hx.send_off[k]; m < hx.send_off[k + 1]; m++) {
                    const int32_t at = fill[static_cast<size_t>(sidx[static_cast<size_t>(m)])]++;
                    peer[static_cast<size_t>(at)] = static_cast<int8_t>(k);
                    tgt[static_cast<size_t>(at)] = target[static_cast<size_t>(m)];
                }
            for (void *old : {static_cast<void *>(hx.node_send_ptr), static_cast<void *>(hx.node_send_target), static_cast<void *>(hx.node_send_peer)}) if (old) HIP_CHECK(hipFree(old));
            hx.node_send_ptr = dev_upload(ptr);
            if (tgt.empty()) { tgt.push_back(0); peer.push_back(0); }
            hx.node_send_target = dev_upload(tgt);
            hx.node_send_peer = dev_upload(peer);
        }
        hx.ipc = true;
    });
}

// back to the buffered form (RCCL send / receive): the neighbours' mappings are closed.  Synchronises.
int mgcfd_rank_ipc_detach(mgcfd_solver *s, int level)
{
    REQUIRE(s);
    return guarded([&] {
        s->use_device();
        DeviceLevel &lv = s->level(level);
        if (!lv.hx || !lv.hx->ipc) return;
        HaloExchange &hx = *lv.hx;
        HIP_CHECK(hipStreamSynchronize(s->stream));
        for (void *m : hx.ipc_opened) (void)hipIpcCloseMemHandle(m);
        hx.ipc_opened.clear();
        hx.ipc = false;
        hx.ipc_all = false;
        hx.ipc_unacknowledged = false;
        if (hx.ipc_timeouts) HIP_CHECK(hipMemset(hx.ipc_timeouts, 0, sizeof(int)));
    });
}

// how many waits for a neighbour's message gave up since the last call (0: every message arrived).  Synchronises.
int mgcfd_rank_ipc_status(mgcfd_solver *s, int level, int *timed_out)
{
    REQUIRE(s); REQUIRE(timed_out);
    return guarded([&] {
        s->use_device();
        DeviceLevel &lv = s->level(level);
        *timed_out = 0;
        if (!lv.hx || !lv.hx->ipc) return;
        HIP_CHECK(hipMemcpyAsync(timed_out, lv.hx->ipc_timeouts, sizeof(int), hipMemcpyDeviceToHost, s->stream));
        HIP_CHECK(hipMemsetAsync(lv.hx->ipc_timeouts, 0, sizeof(int), s->stream));
        HIP_CHECK(hipStreamSynchronize(s->stream));
        lv.hx->ipc_unacknowledged = false;
    });
}

int mgcfd_rank_detach(mgcfd_solver *s)
{
    REQUIRE(s);
    return guarded([&] {
        auto it = g_comms.find(s);
        if (it == g_comms.end()) return;
        if (it->second.rccl) { s->use_device(); (void)hipStreamSynchronize(s->stream); (void)g_rccl.CommDestroy(it->second.rccl); }
        g_comms.erase(it);
    });
}

int mgcfd_rank_set_halo(mgcfd_solver *s, int level, int n_peers, const int *peers, const int64_t *send_counts, const int64_t *const *send_ids,
                        const int64_t *recv_counts, const int64_t *const *recv_ids)
{
    REQUIRE(s);
    if (n_peers > 0) { REQUIRE(peers); REQUIRE(send_counts); REQUIRE(send_ids); REQUIRE(recv_counts); REQUIRE(recv_ids); }
    return guarded([&] { build_halo(s, level, n_peers, peers, send_counts, send_ids, recv_counts, recv_ids, comm_of(s).world); });
}

// what a replayed sweep leaves behind on the host side (the captured launches carry their arguments; the flags and
// counters the eager path updates as it goes are advanced here)
static void after_replayed_sweep(mgcfd_solver *s, int level, const int64_t *iters_delta)
{
    DeviceLevel &lv = s->level(level);
    lv.rot = (lv.rot + 1) % 3;
    lv.apply_rot();
    lv.stage_out = lv.q;
    lv.min_ahead = s->mesh_variant != MGCFD_MESH_FVCORR && part_look_ahead();    // (as stage_interior leaves it)
    for (int k = 0; k < MGCFD_NUM_LOOPS; k++) lv.iters[k] += iters_delta[k];
}

// (MGCFD_SWEEP_GRAPH=0 switches every captured partitioned sweep off, whatever the solvers' MGCFD_OPT_GRAPH says)
static bool sweep_graphs_enabled()
{
    static const bool on = !(std::getenv("MGCFD_SWEEP_GRAPH") && std::atoi(std::getenv("MGCFD_SWEEP_GRAPH")) == 0);
    return on;
}

// Bring the ghosts of `variables` up to date (after mgcfd_set_array, before the first sweep).  In an in-process group
// call mgcfd_group_exchange instead.
int mgcfd_rank_exchange(mgcfd_solver *s, int level)
{
    OP({
        DeviceLevel &lv = s->level(level);
        if (!lv.hx) throw std::invalid_argument("the level has no halo lists");
        if (lv.hx->ipc) {
            // direct mode between processes: behind whatever the peers still read of the last sweep, this rank's owned
            // `variables` into their ghost slots; the rank's own stream goes on behind the messages INTO it
            s->settle_residuals(lv);
            ipc_wait(s, *lv.hx, 2);
            ipc_push(s, lv, lv.q, 0, 2);
            ipc_wait(s, *lv.hx, 2);
            lv.min_ahead = false;
            return;
        }
        if (!comm_of(s).rccl) throw std::invalid_argument("in-process ranks exchange through mgcfd_group_exchange");
        halo_start(s, level, lv.q, 0);
        halo_finish(s, level, lv.q, 0);
        lv.min_ahead = false;
    });
}

int mgcfd_rank_sweeps(mgcfd_solver *s, int level, int sweeps)
{
    OP({
        mgcfd_comm &c = comm_of(s);
        DeviceLevel &lv = s->level(level);
        if (!lv.hx) throw std::invalid_argument("the level has no halo lists: call mgcfd_rank_set_halo first");
        HaloExchange &hx = *lv.hx;
        s->settle_fluxes(lv);                               // (outside any capture: a lazy zero is not part of a sweep)
        if (hx.ipc) {
            if (hx.ipc_unacknowledged) throw HipError("a wait for a neighbouring rank's message gave up earlier and nobody has asked mgcfd_rank_ipc_status since: no further sweeps on stale ghosts");
            for (int k = 0; k < sweeps; k++) rank_sweep_once_ipc(s, level);
            ipc_wait(s, hx, 2);                             // (the ghosts of `variables` are complete in this rank's stream order)
            return;
        }
        if (!c.rccl) throw std::invalid_argument("in-process ranks sweep through mgcfd_group_sweeps");
        for (int k = 0; k < sweeps; k++) {
            const int rot = lv.rot % 3;
            // MGCFD_OPT_GRAPH = 1: the sweep replayed from a hipGraph (measured with the one rank a one-GPU box offers: 76 us
            // per sweep against 129 us issued call by call; ncclSend/ncclRecv inside a capture could not be rehearsed
            // there, hence opt-in)
            // (global time step: a sweep is captured / replayed only from a looked-ahead state — the steady state; the first
            //  sweep after an exchange, which still runs k_step_factor_local, goes out eagerly)
            const bool steady = s->mesh_variant == MGCFD_MESH_FVCORR || lv.min_ahead || !part_look_ahead();
            if (s->opt_graph && sweep_graphs_enabled() && !hx.graph_failed && s->opt_timing == 0 && steady) {
                if (!hx.sweep_graph[rot]) {
                    const bool ahead_before = lv.min_ahead;
                    int64_t before[MGCFD_NUM_LOOPS];
                    std::memcpy(before, lv.iters, sizeof(before));
                    const int rot_before = lv.rot;
                    const bool ok = capture_sweep(s->stream, {}, hx.joined, [&] { rank_sweep_once(s, level); }, &hx.sweep_graph[rot]);
                    s->force_check = -1;
                    if (!ok) {                              // nothing ran: put the host state back and go on eagerly
                        hx.graph_failed = true;
                        lv.rot = rot_before; lv.apply_rot();
                        lv.min_ahead = ahead_before;
                        std::memcpy(lv.iters, before, sizeof(before));
                        rank_sweep_once(s, level);
                        continue;
                    }
                    for (int q = 0; q < MGCFD_NUM_LOOPS; q++) { hx.graph_iters[rot][q] = lv.iters[q] - before[q]; lv.iters[q] = before[q]; }
                    lv.rot = rot_before; lv.apply_rot();    // (the capture advanced the host state without running anything)
                    lv.min_ahead = ahead_before;
                }
                HIP_CHECK(hipGraphLaunch(hx.sweep_graph[rot], s->stream));
                hx.sweeps_replayed++;
                after_replayed_sweep(s, level, hx.graph_iters[rot]);
            } else {
                rank_sweep_once(s, level);
            }
        }
    });
}

// calc_rms over a partitioned level (src/Kernels/validation.cpp:91-105): the sum of squared residuals over the OWNED nodes
// of every rank (all-reduce SUM of one fp64); rms = sqrt(sum / nodes of the whole level).  Synchronises.
int mgcfd_rank_residual_sumsq(mgcfd_solver *s, int level, double *sum_all_ranks)
{
    REQUIRE(sum_all_ranks);
    OP({
        mgcfd_comm &c = comm_of(s);
        if (!c.rccl) throw std::invalid_argument("in-process ranks: mgcfd_group_rms");
        DeviceLevel &lv = s->level(level);
        if (!lv.hx) throw std::invalid_argument("the level has no halo lists");
        exact::launch_sumsq(s->stream, lv.info.nel, lv.dp.stride, lv.residuals, lv.partials, lv.n_partials, lv.sumsq, lv.dp.old_of_new, lv.n_owned);
        RCCL_CHECK(g_rccl.AllReduce(lv.sumsq, lv.hx->gmin, 1, Rccl::kDouble, Rccl::kSum, c.rccl, s->stream));
        HIP_CHECK(hipMemcpyAsync(sum_all_ranks, lv.hx->gmin, sizeof(double), hipMemcpyDeviceToHost, s->stream));
        HIP_CHECK(hipStreamSynchronize(s->stream));
    });
}

// ---- in-process groups: the ranks are solvers of this process (one per device — or several on one device in tests) ----
int mgcfd_group_create(int n, mgcfd_solver *const *solvers, mgcfd_group **out)
{
    REQUIRE(solvers); REQUIRE(out);
    return guarded([&] {
        if (n < 1) throw std::invalid_argument("a group needs at least one rank");
        auto g = std::make_unique<mgcfd_group>();
        for (int r = 0; r < n; r++) {
            if (!solvers[r]) throw std::invalid_argument("null solver");
            g->ranks.push_back(solvers[r]);
        }
        // peer access between the devices of the group (xGMI): a copy between two devices then goes direct
        for (int a = 0; a < n; a++)
            for (int b = 0; b < n; b++) {
                if (solvers[a]->device == solvers[b]->device) continue;
                int can = 0;
                HIP_CHECK(hipDeviceCanAccessPeer(&can, solvers[a]->device, solvers[b]->device));
                if (can) { solvers[a]->use_device(); const hipError_t e = hipDeviceEnablePeerAccess(solvers[b]->device, 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_CHECK(e); (void)hipGetLastError(); }
                else g->peer_ok = false;
            }
        // the group's all-reduce reads the other ranks' scalars in place and its messages are stores into the peers' memory:
        // a kernel must never touch memory its device cannot address, so such a set of devices is refused here
        if (!g->peer_ok) throw std::invalid_argument("the devices of an in-process group must have peer access to each other (use one rank per process over RCCL instead)");
        for (int r = 0; r < n; r++) {
            mgcfd_comm c;
            c.rank = r; c.world = n; c.group = g.get();
            g_comms[solvers[r]] = c;
        }
        *out = g.release();
    });
}

void mgcfd_group_destroy(mgcfd_group *g)
{
    if (!g) return;
    for (mgcfd_solver *s : g->ranks) g_comms.erase(s);
    delete g;
}

int mgcfd_group_exchange(mgcfd_group *g, int level)
{
    REQUIRE(g);
    return guarded([&] {
        for (mgcfd_solver *s : g->ranks) if (!s->level(level).hx) throw std::invalid_argument("a rank has no halo lists");
        group_prepare_direct(g, level);
        if (g->ranks[0]->level(level).hx->direct) {
            // every rank's owned `variables` into its peers' ghost slots; a peer's stream continues behind the pushes into it
            for (mgcfd_solver *s : g->ranks) {
                s->use_device();
                DeviceLevel &lv = s->level(level);
                s->settle_residuals(lv);
                wait_for_peers(g, s, level, 2);             // (nobody still reads the ghosts a peer is about to overwrite: their last sweep's last stage is behind this)
                push_and_record(s, level, lv.q, make_push_peers(g, s, level, [](DeviceLevel &pl) { return pl.q; }), 2);
                lv.min_ahead = false;
            }
            for (mgcfd_solver *s : g->ranks) { s->use_device(); wait_for_peers(g, s, level, 2); }
            return;
        }
        for (mgcfd_solver *s : g->ranks) { s->use_device(); DeviceLevel &lv = s->level(level); halo_start(s, level, lv.q, 0); }
        group_deliver(g, level, 0);
        for (mgcfd_solver *s : g->ranks) { s->use_device(); DeviceLevel &lv = s->level(level); halo_finish(s, level, lv.q, 0); lv.min_ahead = false; }
    });
}

// rms_out (may be null): the RMS after each of the `sweeps` sweeps, read back once at the end (direct mode only)
static int group_sweeps_impl(mgcfd_group *g, int level, int sweeps, double *rms_out)
{
    REQUIRE(g);
    return guarded([&] {
        if (rms_out && sweeps > mgcfd_solver::kRmsRing) throw std::invalid_argument("at most 4096 sweeps per call with the RMS of each");
        mgcfd_solver *s0 = g->ranks[0];
        for (mgcfd_solver *s : g->ranks) {
            s->use_device();
            DeviceLevel &lv = s->level(level);
            if (!lv.hx) throw std::invalid_argument("a rank has no halo lists: call mgcfd_rank_set_halo first");
            s->settle_fluxes(lv);
        }
        for (mgcfd_solver *s : g->ranks) {                  // (allocations and uploads: never inside a capture)
            HaloExchange &hx = *s->level(level).hx;
            if (hx.min_par) continue;
            s->use_device();
            hx.min_par = dev_alloc<double>(2);
        }
        for (mgcfd_solver *s : g->ranks) {
            HaloExchange &hx = *s->level(level).hx;
            if (hx.peer_scalars[0]) continue;
            s->use_device();
            for (int par = 0; par < 2; par++) {
                std::vector<const double *> ptrs;
                for (mgcfd_solver *src : g->ranks) ptrs.push_back(src->level(level).hx->min_par + par);
                hx.peer_scalars[par] = dev_upload(ptrs);
            }
        }
        group_prepare_direct(g, level);
        HaloExchange &h0 = *s0->level(level).hx;
        bool timing = false;
        for (mgcfd_solver *s : g->ranks) timing = timing || s->opt_timing != 0;
        // direct mode: a host thread per rank (MGCFD_GROUP_THREADS=0: one thread issues everything, for A/B)
        const bool threads_wanted = !(std::getenv("MGCFD_GROUP_THREADS") && std::atoi(std::getenv("MGCFD_GROUP_THREADS")) == 0);
        if (rms_out && !h0.direct) {
            // the buffered form: the RMS read back after every sweep (as mgcfd_group_rms does it)
            for (int k = 0; k < sweeps; k++) {
                group_sweep_once(g, level);
                double sum = 0.0;
                int64_t nodes = 0;
                for (mgcfd_solver *s : g->ranks) {
                    s->use_device();
                    DeviceLevel &lv = s->level(level);
                    exact::launch_sumsq(s->stream, lv.info.nel, lv.dp.stride, lv.residuals, lv.partials, lv.n_partials, lv.sumsq, lv.dp.old_of_new, lv.n_owned);
                    double part = 0.0;
                    HIP_CHECK(hipMemcpyAsync(&part, lv.sumsq, sizeof(double), hipMemcpyDeviceToHost, s->stream));
                    HIP_CHECK(hipStreamSynchronize(s->stream));
                    sum += part;
                    nodes += lv.n_owned;
                }
                rms_out[k] = std::sqrt(sum / double(nodes));
            }
            return;
        }
        // (the threads are started per call — ~50 us — so a call of one or two sweeps is issued by the caller's thread alone)
        const bool threaded = threads_wanted && g->ranks.size() > 1 && sweeps >= 4;
        if (h0.direct && (rms_out || threaded)) {
            if (rms_out)
                for (mgcfd_solver *s : g->ranks) {
                    s->use_device();
                    if (!s->rms_ring) { s->rms_ring = dev_alloc<double>(mgcfd_solver::kRmsRing); s->rms_count = dev_alloc<int>(1); }
                    HIP_CHECK(hipMemsetAsync(s->rms_count, 0, sizeof(int), s->stream));
                }
            if (threaded) group_sweeps_threaded(g, level, sweeps, rms_out != nullptr);
            else {
                for (int k = 0; k < sweeps; k++) {
                    group_sweep_once(g, level);
                    if (rms_out)
                        for (mgcfd_solver *s : g->ranks) {
                            s->use_device();
                            DeviceLevel &lv = s->level(level);
                            exact::launch_sumsq(s->stream, lv.info.nel, lv.dp.stride, lv.residuals, lv.partials, lv.n_partials, lv.sumsq, lv.dp.old_of_new, lv.n_owned);
                            exact::launch_append_scalar(s->stream, lv.sumsq, s->rms_ring, s->rms_count, mgcfd_solver::kRmsRing);
                        }
                }
                for (mgcfd_solver *s : g->ranks) { s->use_device(); wait_for_peers(g, s, level, 2); }
            }
            if (rms_out) {
                // one read-back: the ranks' sums of every sweep, added in rank order as mgcfd_group_rms adds them
                std::vector<double> sums(static_cast<size_t>(sweeps), 0.0), part(static_cast<size_t>(sweeps));
                int64_t nodes = 0;
                for (mgcfd_solver *s : g->ranks) {
                    s->use_device();
                    HIP_CHECK(hipMemcpyAsync(part.data(), s->rms_ring, sizeof(double) * static_cast<size_t>(sweeps), hipMemcpyDeviceToHost, s->stream));
                    HIP_CHECK(hipStreamSynchronize(s->stream));
                    for (int k = 0; k < sweeps; k++) sums[static_cast<size_t>(k)] += part[static_cast<size_t>(k)];
                    nodes += s->level(level).n_owned;
                }
                for (int k = 0; k < sweeps; k++) rms_out[k] = std::sqrt(sums[static_cast<size_t>(k)] / double(nodes));
            }
            return;
        }
        // One graph over all ranks' streams replays SLOWER than the eager calls on ROCm 7.2 (2 ranks: 301 against 262 us per
        // sweep; launching a graph of ~40 nodes on 4 streams costs the host 150 us), so groups capture only on request
        // (MGCFD_GROUP_GRAPH=1); a single rank's two-stream sweep does gain (mgcfd_rank_sweeps: 76 against 129 us).
        static const bool group_graphs = std::getenv("MGCFD_GROUP_GRAPH") && std::atoi(std::getenv("MGCFD_GROUP_GRAPH")) != 0;
        const bool graphs = group_graphs && sweep_graphs_enabled() && !timing;
        if (graphs) {
            // a replayed graph is launched on rank 0's stream: it must start behind whatever the other ranks' streams hold
            for (mgcfd_solver *s : g->ranks) {
                if (s == s0) continue;
                s->use_device();
                HIP_CHECK(hipEventRecord(s->level(level).hx->joined, s->stream));
                HIP_CHECK(hipStreamWaitEvent(s0->stream, s->level(level).hx->joined, 0));
            }
        }
        bool replayed = false;
        for (int k = 0; k < sweeps; k++) {
            const int rot = s0->level(level).rot % 3 + 3 * h0.min_parity;     // (the captured launches name the minima's word of this parity)
            const bool steady = s0->mesh_variant == MGCFD_MESH_FVCORR || s0->level(level).min_ahead || !part_look_ahead();   // (as in mgcfd_rank_sweeps)
            if (graphs && !h0.graph_failed && steady) {
                if (!h0.sweep_graph[rot]) {
                    const bool ahead_before = s0->level(level).min_ahead;
                    // every rank's sweep in ONE graph: the other ranks' streams fork from rank 0's and join it at the end
                    std::vector<std::vector<int64_t>> before;
                    std::vector<int> rot_before;
                    std::vector<std::pair<hipStream_t, hipEvent_t>> others;
                    for (mgcfd_solver *s : g->ranks) {
                        DeviceLevel &lv = s->level(level);
                        before.emplace_back(lv.iters, lv.iters + MGCFD_NUM_LOOPS);
                        rot_before.push_back(lv.rot);
                        if (s != s0) others.emplace_back(s->stream, lv.hx->joined);
                    }
                    s0->use_device();
                    const bool ok = capture_sweep(s0->stream, others, h0.joined, [&] { group_sweep_once(g, level); }, &h0.sweep_graph[rot]);
                    for (size_t r = 0; r < g->ranks.size(); r++) {
                        mgcfd_solver *s = g->ranks[r];
                        DeviceLevel &lv = s->level(level);
                        s->force_check = -1;
                        for (int q = 0; q < MGCFD_NUM_LOOPS; q++) { lv.hx->graph_iters[rot][q] = lv.iters[q] - before[r][static_cast<size_t>(q)]; lv.iters[q] = before[r][static_cast<size_t>(q)]; }
                        lv.rot = rot_before[r]; lv.apply_rot();
                        lv.hx->min_parity = rot / 3;        // (the capture advanced the host state without running anything)
                        lv.min_ahead = ahead_before;
                    }
                    if (!ok) { h0.graph_failed = true; group_sweep_once(g, level); continue; }
                }
                s0->use_device();
                HIP_CHECK(hipGraphLaunch(h0.sweep_graph[rot], s0->stream));
                replayed = true;
                for (mgcfd_solver *s : g->ranks) { after_replayed_sweep(s, level, s->level(level).hx->graph_iters[rot]); s->level(level).hx->min_parity ^= 1; }
            } else {
                group_sweep_once(g, level);
            }
        }
        // direct mode: the last stage's pushes land in the peers' `variables`; a rank's stream goes on only behind the pushes into it
        if (h0.direct) for (mgcfd_solver *s : g->ranks) { s->use_device(); wait_for_peers(g, s, level, 2); }
        if (replayed) {
            // ... and what the other ranks' streams are given next must wait for the graphs
            s0->use_device();
            HIP_CHECK(hipEventRecord(h0.joined, s0->stream));
            for (mgcfd_solver *s : g->ranks) if (s != s0) { s->use_device(); HIP_CHECK(hipStreamWaitEvent(s->stream, h0.joined, 0)); }
        }
    });
}

int mgcfd_group_sweeps(mgcfd_group *g, int level, int sweeps) { return group_sweeps_impl(g, level, sweeps, nullptr); }
int mgcfd_group_sweeps_rms(mgcfd_group *g, int level, int sweeps, double *rms_of_each)
{
    REQUIRE(rms_of_each);
    return group_sweeps_impl(g, level, sweeps, rms_of_each);
}

// calc_rms over the whole level: sqrt(sum over the ranks' owned nodes of residual^2 / owned nodes of all ranks).  Synchronises.
int mgcfd_group_rms(mgcfd_group *g, int level, double *rms)
{
    REQUIRE(g); REQUIRE(rms);
    return guarded([&] {
        double sum = 0.0;
        int64_t nodes = 0;
        for (mgcfd_solver *s : g->ranks) {
            s->use_device();
            DeviceLevel &lv = s->level(level);
            exact::launch_sumsq(s->stream, lv.info.nel, lv.dp.stride, lv.residuals, lv.partials, lv.n_partials, lv.sumsq, lv.dp.old_of_new, lv.n_owned);
            double part = 0.0;
            HIP_CHECK(hipMemcpyAsync(&part, lv.sumsq, sizeof(double), hipMemcpyDeviceToHost, s->stream));
            HIP_CHECK(hipStreamSynchronize(s->stream));
            sum += part;
            nodes += lv.n_owned;
        }
        *rms = std::sqrt(sum / double(nodes));
    });
}

int mgcfd_group_synchronize(mgcfd_group *g)
{
    REQUIRE(g);
    return guarded([&] { for (mgcfd_solver *s : g->ranks) { s->use_device(); HIP_CHECK(hipStreamSynchronize(s->stream)); HIP_CHECK(hipGetLastError()); } });
}

// ---- V-cycles on a PARTITIONED hierarchy, the whole state machine inside the library --------------------------------------
// mgcfd_create_partitioned_mg + mgcfd_rank_set_halo on EVERY level.  The cycle is the reference's
// (src/euler3d_cpu_double.cpp:371-694: sweeps on levels 0 .. n-1, n-2 .. 1 with mg_restrict on the way up and
// prolong_residuals_interpolate_proper on the way down) with ghost values moved where the next loop reads them:
//   `variables` of a level   after every time_step (inside the partitioned sweep: one message per Runge-Kutta stage),
//                            after mg_restrict filled it (coarse ghosts: mg_loops.cpp:30-202 writes owned coarse nodes only) and
//                            after the prolongation corrected it (fine ghosts: the next sweep's fluxes read them);
//   `residuals` of the coarse level before the prolongation (mg_loops.cpp:678-864 reads the parents of a node's neighbours).
// A coarse node is averaged where it is owned, over its children in GLOBAL-id order (order_keys), so every level equals the
// unpartitioned hierarchy's bit for bit on owned nodes.  The transfers' exchanges are the buffered form (pack, copies or RCCL
// send / receive, unpack) outside the sweeps' three-set rotation, hence with explicit waits for the buffers.

static void group_prepare_level(mgcfd_group *g, int level)
{
    for (mgcfd_solver *s : g->ranks) {
        s->use_device();
        DeviceLevel &lv = s->level(level);
        if (!lv.hx) throw std::invalid_argument("a rank has no halo lists on level " + std::to_string(level) + ": call mgcfd_rank_set_halo for every level");
        s->settle_fluxes(lv);
    }
    for (mgcfd_solver *s : g->ranks) {                      // (allocations and uploads: never inside a sweep)
        HaloExchange &hx = *s->level(level).hx;
        if (hx.min_par) continue;
        s->use_device();
        hx.min_par = dev_alloc<double>(2);
    }
    for (mgcfd_solver *s : g->ranks) {
        HaloExchange &hx = *s->level(level).hx;
        if (hx.peer_scalars[0]) continue;
        s->use_device();
        for (int par = 0; par < 2; par++) {
            std::vector<const double *> ptrs;
            for (mgcfd_solver *src : g->ranks) ptrs.push_back(src->level(level).hx->min_par + par);
            hx.peer_scalars[par] = dev_upload(ptrs);
        }
    }
    group_prepare_direct(g, level);
}

static double *array_ptr(DeviceLevel &lv, int which, int *ncols);

// the ghosts of array `which` of `level` <- their owners, on every rank of the group
static void group_exchange_array(mgcfd_group *g, int level, int which)
{
    const int b = 0;
    for (mgcfd_solver *s : g->ranks) {
        s->use_device();
        DeviceLevel &lv = s->level(level);
        HaloExchange &hx = *lv.hx;
        s->settle_residuals(lv);
        // this rank's send buffer is free behind the copies the peers made out of it (their `arrived` of the last use of the
        // set); and in direct mode nobody's last push into this rank's ghosts may still be on its way when the unpack writes them
        for (int p : hx.peer) HIP_CHECK(hipStreamWaitEvent(s->stream, g->ranks[static_cast<size_t>(p)]->level(level).hx->arrived[b], 0));
        if (hx.direct) wait_for_peers(g, s, level, 2);
        int nc = 0;
        halo_start(s, level, array_ptr(lv, which, &nc), b);
    }
    group_deliver(g, level, b, true);
    for (mgcfd_solver *s : g->ranks) {
        s->use_device();
        DeviceLevel &lv = s->level(level);
        int nc = 0;
        halo_finish(s, level, array_ptr(lv, which, &nc), b);
        if (which == MGCFD_ARR_VARIABLES) lv.min_ahead = false;
    }
}

static void append_level0_sumsq(mgcfd_solver *s)
{
    DeviceLevel &lv = s->level(0);
    exact::launch_sumsq(s->stream, lv.info.nel, lv.dp.stride, lv.residuals, lv.partials, lv.n_partials, lv.sumsq, lv.dp.old_of_new, lv.n_owned);
    exact::launch_append_scalar(s->stream, lv.sumsq, s->rms_ring, s->rms_count, mgcfd_solver::kRmsRing);
}

// one V-cycle of every rank of an in-process group (one host thread issues it)
static void group_cycle_once(mgcfd_group *g, bool with_rms)
{
    const int n = static_cast<int>(g->ranks[0]->L.size());
    for (int l = 0; l < n; l++) {
        group_sweep_once(g, l);                                                    // :383-508
        if (l == 0 && with_rms) for (mgcfd_solver *s : g->ranks) { s->use_device(); append_level0_sumsq(s); }   // :509-512
        if (l + 1 < n) {
            for (mgcfd_solver *s : g->ranks) {
                s->use_device();
                if (s->level(l).hx->direct) wait_for_peers(g, s, l, 2);            // (the children this rank holds as ghosts: the peers' last pushes)
                s->op_restrict(l);                                                 // :527-559
            }
            group_exchange_array(g, l + 1, MGCFD_ARR_VARIABLES);
        }
    }
    for (int l = n - 2; l >= 0; l--) {
        group_exchange_array(g, l + 1, MGCFD_ARR_RESIDUALS);
        for (mgcfd_solver *s : g->ranks) { s->use_device(); s->op_prolong(l); }    // :560-688
        group_exchange_array(g, l, MGCFD_ARR_VARIABLES);
        if (l > 0) group_sweep_once(g, l);
    }
}

// The same cycle with a host thread per rank (direct mode; what mgcfd_group_cycles runs for groups of several ranks:
// one thread issuing every rank's ~100 calls per cycle makes the host the bottleneck N times over).  As in
// group_sweeps_threaded the threads agree on the ORDER of event records and waits through barriers: one after the first
// half of compute_step_factor and after every stage of a sweep, two per transfer exchange (all packed | all delivered).
static void group_cycles_threaded(mgcfd_group *g, int cycles, bool with_rms)
{
    const int n = static_cast<int>(g->ranks.size());
    const int nl = static_cast<int>(g->ranks[0]->L.size());
    SpinBarrier bar(n);
    std::atomic<bool> failed{false};
    std::mutex mu;
    std::exception_ptr first_error;
    const bool global_dt = g->ranks[0]->mesh_variant != MGCFD_MESH_FVCORR;
    auto run = [&](int r) {
        mgcfd_solver *s = g->ranks[static_cast<size_t>(r)];
        // (a rank that failed keeps arriving at the barriers, doing nothing, so that the others are not left waiting)
        auto step = [&](auto &&body) {
            if (failed.load(std::memory_order_acquire)) return;
            try { body(); }
            catch (...) { std::lock_guard<std::mutex> lock(mu); if (!first_error) first_error = std::current_exception(); failed.store(true, std::memory_order_release); }
        };
        step([&] { s->use_device(); });
        auto sweep = [&](int level) {                   // (group_sweeps_threaded's body)
            DeviceLevel &lv = s->level(level);
            HaloExchange &hx = *lv.hx;
            const int par = hx.min_parity;
            step([&] {
                sweep_first_half(s, level, hx.min_par + par);
                if (global_dt) HIP_CHECK(hipEventRecord(hx.reduced, s->stream));
            });
            hx.min_parity = par ^ 1;
            bar.wait();
            PushPeers to[MGCFD_RK];
            step([&] {
                if (global_dt) {
                    for (mgcfd_solver *src : g->ranks) if (src != s) HIP_CHECK(hipStreamWaitEvent(s->stream, src->level(level).hx->reduced, 0));
                    exact::launch_min_over_peers(s->stream, hx.peer_scalars[par], n, hx.gmin);
                }
                for (int j = 0; j < MGCFD_RK; j++) to[j] = make_push_peers(g, s, level, [&](DeviceLevel &pl) { return stage_out_buffer(pl, j); });
            });
            for (int j = 0; j < MGCFD_RK; j++) {
                step([&] {
                    stage_boundary_direct(g, s, level, j, global_dt ? 3 : 0, hx.gmin, 1, &to[j]);
                    stage_interior(s, level, j, global_dt ? 3 : 0, hx.gmin, 1);
                });
                bar.wait();
            }
        };
        auto exchange = [&](int level, int which) {
            const int b = 0;
            DeviceLevel &lv = s->level(level);
            HaloExchange &hx = *lv.hx;
            step([&] {
                s->settle_residuals(lv);
                for (int p : hx.peer) HIP_CHECK(hipStreamWaitEvent(s->stream, g->ranks[static_cast<size_t>(p)]->level(level).hx->arrived[b], 0));
                wait_for_peers(g, s, level, 2);
                int nc = 0;
                halo_start(s, level, array_ptr(lv, which, &nc), b);
            });
            bar.wait();                                 // every rank's pack event is recorded
            step([&] {
                deliver_to(g, s, level, b, true);
                int nc = 0;
                halo_finish(s, level, array_ptr(lv, which, &nc), b);
                if (which == MGCFD_ARR_VARIABLES) lv.min_ahead = false;
            });
            bar.wait();                                 // ... and every rank's arrival event, before anybody's next wait for it
        };
        for (int c = 0; c < cycles; c++) {
            for (int l = 0; l < nl; l++) {
                sweep(l);
                if (l == 0 && with_rms) step([&] { append_level0_sumsq(s); });
                if (l + 1 < nl) {
                    step([&] { wait_for_peers(g, s, l, 2); s->op_restrict(l); });
                    exchange(l + 1, MGCFD_ARR_VARIABLES);
                }
            }
            for (int l = nl - 2; l >= 0; l--) {
                exchange(l + 1, MGCFD_ARR_RESIDUALS);
                step([&] { s->op_prolong(l); });
                exchange(l, MGCFD_ARR_VARIABLES);
                if (l > 0) sweep(l);
            }
        }
        step([&] { for (int l = 0; l < nl; l++) wait_for_peers(g, s, l, 2); HIP_CHECK(hipGetLastError()); });
    };
    std::vector<std::thread> threads;
    for (int r = 1; r < n; r++) threads.emplace_back(run, r);
    run(0);
    for (std::thread &t : threads) t.join();
    if (first_error) std::rethrow_exception(first_error);
}

// what the checks inside the launches of every rank found (the reference exits at the first bad cell of the first failing time_step)
static int group_read_errors(mgcfd_group *g)
{
    int code = MGCFD_OK;
    for (mgcfd_solver *s : g->ranks) {
        s->use_device();
        const int c = s->read_error(nullptr);
        if (c != MGCFD_OK && code == MGCFD_OK) code = c;
    }
    return code;
}

int mgcfd_group_cycles(mgcfd_group *g, int cycles, double *rms_out)
{
    REQUIRE(g);
    int code = MGCFD_OK;
    const int rc = guarded([&] {
        if (cycles > mgcfd_solver::kRmsRing) throw std::invalid_argument("at most 4096 cycles per call");
        const int n = static_cast<int>(g->ranks[0]->L.size());
        for (mgcfd_solver *s : g->ranks) if (static_cast<int>(s->L.size()) != n) throw std::invalid_argument("the ranks of a group hold the same number of levels");
        for (int l = 0; l < n; l++) group_prepare_level(g, l);
        for (mgcfd_solver *s : g->ranks) {
            s->use_device();
            if (!s->rms_ring) { s->rms_ring = dev_alloc<double>(mgcfd_solver::kRmsRing); s->rms_count = dev_alloc<int>(1); }
            HIP_CHECK(hipMemsetAsync(s->rms_count, 0, sizeof(int), s->stream));
        }
        // A host thread per rank where every level runs the direct form and every rank has a device of its own: one thread
        // issuing N ranks' ~100 calls per cycle costs N x 0.76 ms (tools/hostcost_cycles.py).  Ranks that SHARE a device (the
        // one-GPU rehearsal) are issued by the caller's thread: there the threads only contend for the one device's queues
        // (8 ranks: 12.2 ms per cycle against 6.0).  MGCFD_GROUP_THREADS=0 / 1 forces either form.
        bool all_direct = g->ranks.size() > 1;
        for (int l = 0; l < n; l++) all_direct = all_direct && g->ranks[0]->level(l).hx->direct;
        if (const char *e = std::getenv("MGCFD_GROUP_THREADS")) all_direct = all_direct && std::atoi(e) != 0;
        else {
            std::vector<int> devs;
            for (mgcfd_solver *s : g->ranks) devs.push_back(s->device);
            std::sort(devs.begin(), devs.end());
            all_direct = all_direct && std::adjacent_find(devs.begin(), devs.end()) == devs.end();
        }
        if (all_direct) group_cycles_threaded(g, cycles, rms_out != nullptr);
        else for (int c = 0; c < cycles; c++) group_cycle_once(g, rms_out != nullptr);
        // every rank's stream behind the last pushes into it, then the read-backs
        for (mgcfd_solver *s : g->ranks) { s->use_device(); for (int l = 0; l < n; l++) if (s->level(l).hx->direct) wait_for_peers(g, s, l, 2); }
        if (rms_out) {
            std::vector<double> sums(static_cast<size_t>(cycles), 0.0), part(static_cast<size_t>(std::max(cycles, 1)));
            int64_t nodes = 0;
            for (mgcfd_solver *s : g->ranks) {                  // (added in rank order, as mgcfd_group_rms adds them)
                s->use_device();
                HIP_CHECK(hipMemcpyAsync(part.data(), s->rms_ring, sizeof(double) * static_cast<size_t>(cycles), hipMemcpyDeviceToHost, s->stream));
                HIP_CHECK(hipStreamSynchronize(s->stream));
                for (int k = 0; k < cycles; k++) sums[static_cast<size_t>(k)] += part[static_cast<size_t>(k)];
                nodes += s->level(0).n_owned;
            }
            for (int k = 0; k < cycles; k++) rms_out[k] = std::sqrt(sums[static_cast<size_t>(k)] / double(nodes));
        }
        code = group_read_errors(g);
        for (mgcfd_solver *s : g->ranks) { s->use_device(); HIP_CHECK(hipGetLastError()); }
    });
    if (rc != MGCFD_OK) return rc;
    if (code != MGCFD_OK) g_last_error = "check_for_invalid_variables: a rank of the group found an invalid state during the cycles";
    return code;
}

// ---- the same for one rank per process over RCCL ----
static void rank_exchange_array(mgcfd_solver *s, int level, int which)
{
    DeviceLevel &lv = s->level(level);
    s->settle_residuals(lv);
    int nc = 0;
    double *field = array_ptr(lv, which, &nc);
    halo_start(s, level, field, 0);                         // (set 0 is free: a sweep finishes every exchange it starts, and so does this)
    halo_finish(s, level, field, 0);
    if (which == MGCFD_ARR_VARIABLES) lv.min_ahead = false;
}

static void rank_cycle_once(mgcfd_solver *s, bool with_rms)
{
    const int n = static_cast<int>(s->L.size());
    for (int l = 0; l < n; l++) {
        rank_sweep_once(s, l);
        if (l == 0 && with_rms) append_level0_sumsq(s);
        if (l + 1 < n) { s->op_restrict(l); rank_exchange_array(s, l + 1, MGCFD_ARR_VARIABLES); }
    }
    for (int l = n - 2; l >= 0; l--) {
        rank_exchange_array(s, l + 1, MGCFD_ARR_RESIDUALS);
        s->op_prolong(l);
        rank_exchange_array(s, l, MGCFD_ARR_VARIABLES);
        if (l > 0) rank_sweep_once(s, l);
    }
}

int mgcfd_rank_cycles(mgcfd_solver *s, int cycles, double *rms_out)
{
    REQUIRE(s);
    int code = MGCFD_OK;
    const int rc = guarded([&] {
        s->use_device();
        mgcfd_comm &c = comm_of(s);
        if (!c.rccl) throw std::invalid_argument("in-process ranks cycle through mgcfd_group_cycles");
        if (cycles > mgcfd_solver::kRmsRing) throw std::invalid_argument("at most 4096 cycles per call");
        for (DeviceLevel &lv : s->L) {
            if (!lv.hx) throw std::invalid_argument("a level has no halo lists: call mgcfd_rank_set_halo for every level");
            if (lv.hx->ipc) throw std::invalid_argument("mgcfd_rank_cycles runs the buffered exchange: mgcfd_rank_ipc_detach first");
            s->settle_fluxes(lv);
        }
        if (!s->rms_ring) { s->rms_ring = dev_alloc<double>(mgcfd_solver::kRmsRing); s->rms_count = dev_alloc<int>(1); }
        HIP_CHECK(hipMemsetAsync(s->rms_count, 0, sizeof(int), s->stream));
        for (int k = 0; k < cycles; k++) rank_cycle_once(s, rms_out != nullptr);
        if (rms_out && cycles > 0) {
            // calc_rms of the whole level (validation.cpp:91-105): the ranks' sums of every cycle added by ONE all-reduce
            RCCL_CHECK(g_rccl.AllReduce(s->rms_ring, s->rms_ring, static_cast<size_t>(cycles), Rccl::kDouble, Rccl::kSum, c.rccl, s->stream));
            double nodes_local = double(s->level(0).n_owned);
            DeviceLevel &l0 = s->level(0);
            HIP_CHECK(hipMemcpyAsync(l0.sumsq, &nodes_local, sizeof(double), hipMemcpyHostToDevice, s->stream));
            RCCL_CHECK(g_rccl.AllReduce(l0.sumsq, l0.hx->gmin, 1, Rccl::kDouble, Rccl::kSum, c.rccl, s->stream));
            double nodes = 0.0;
            std::vector<double> sums(static_cast<size_t>(cycles));
            HIP_CHECK(hipMemcpyAsync(sums.data(), s->rms_ring, sizeof(double) * static_cast<size_t>(cycles), hipMemcpyDeviceToHost, s->stream));
            HIP_CHECK(hipMemcpyAsync(&nodes, l0.hx->gmin, sizeof(double), hipMemcpyDeviceToHost, s->stream));
            HIP_CHECK(hipStreamSynchronize(s->stream));
            for (int k = 0; k < cycles; k++) rms_out[k] = std::sqrt(sums[static_cast<size_t>(k)] / nodes);
        }
        code = s->read_error(nullptr);                      // synchronises
        HIP_CHECK(hipGetLastError());
    });
    if (rc != MGCFD_OK) return rc;
    if (code != MGCFD_OK) g_last_error = "check_for_invalid_variables: invalid state during the cycles";
    return code;
}

// What a solver is a rank of, AS THE LIBRARY SEES IT: out[0] this rank, out[1] the number of ranks, out[2] the transport
// (0 none, 1 RCCL communicator, 2 in-process group, 3 plain attachment: messages through HIP IPC only), out[3] the size the
// RCCL communicator itself reports (ncclCommCount; -1 when there is none).
int mgcfd_rank_info(const mgcfd_solver *s, int out[4])
{
    REQUIRE(s); REQUIRE(out);
    out[0] = 0; out[1] = 1; out[2] = 0; out[3] = -1;
    const auto it = g_comms.find(const_cast<mgcfd_solver *>(s));
    if (it == g_comms.end()) return MGCFD_OK;
    const mgcfd_comm &c = it->second;
    out[0] = c.rank; out[1] = c.world; out[2] = c.rccl ? 1 : (c.group ? 2 : 3);
    if (c.rccl && g_rccl.CommCount) { int n = -1; if (g_rccl.CommCount(c.rccl, &n) == 0) out[3] = n; }
    return MGCFD_OK;
}

int mgcfd_rank_graph_status(const mgcfd_solver *s, int level, int64_t out[3])
{
    REQUIRE(s); REQUIRE(out);
    if (level < 0 || level >= static_cast<int>(s->L.size()) || !s->L[static_cast<size_t>(level)].hx) { g_last_error = "the level has no halo lists"; return MGCFD_ERR_ARG; }
    const HaloExchange &hx = *s->L[static_cast<size_t>(level)].hx;
    out[0] = 0;
    for (hipGraphExec_t ge : hx.sweep_graph) if (ge) out[0]++;
    out[1] = hx.graph_failed ? 1 : 0;
    out[2] = hx.sweeps_replayed;
    return MGCFD_OK;
}

// how a level's tiles split for the overlapped exchange: out[0] boundary tiles, out[1] interior tiles, out[2] nodes sent, out[3] nodes received
int mgcfd_rank_halo_info(const mgcfd_solver *s, int level, int64_t out[4])
{
    REQUIRE(s); REQUIRE(out);
    if (level < 0 || level >= static_cast<int>(s->L.size()) || !s->L[static_cast<size_t>(level)].hx) { g_last_error = "the level has no halo lists"; return MGCFD_ERR_ARG; }
    const HaloExchange &hx = *s->L[static_cast<size_t>(level)].hx;
    out[0] = hx.n_boundary; out[1] = hx.n_interior; out[2] = hx.total_send(); out[3] = hx.total_recv();
    return MGCFD_OK;
}

} // extern "C"
