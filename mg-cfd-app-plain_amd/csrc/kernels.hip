// kernels.hip — hand-written HIP kernels for gfx950 (MI355X) for the MG-CFD hot path.
//
// Compiled TWICE into two namespaces:
//   -DMGCFD_KERNEL_NS=exact -ffp-contract=off   every fp64 op rounds as in the reference built
//                                               with -ffp-contract=off; together with the
//                                               reference-order gathers this is bit-identical
//   -DMGCFD_KERNEL_NS=fast  -ffp-contract=fast  same code, FMA contraction allowed
//
// Design: there is no dense contraction here, so no MFMA; every loop is a
// node-centred sweep with one lane per node, over structure-of-arrays state
// (field f of node i at q[f*stride + i]) so that per-node accesses coalesce perfectly.
//   * flux_tile     replaces the reference's edge loop + scatter-add
//                   (src/Kernels/flux_loops.cpp:133-136, flux_kernel.elemfunc.c) by a per-node
//                   GATHER: a 256-thread workgroup owns a compact cluster of 256 nodes, stages
//                   their state and their halo's state into an LDS tile (each node fetched once
//                   per tile instead of once per incident edge), and every lane then walks its
//                   incidence rows (sliced ELLPACK, 26-34 B per entry, coalesced) and sums the
//                   edge fluxes in the reference's order — no atomics, no colouring,
//                   deterministic, bit-identical.
//   * The division / square-root work (8 div + 5 sqrt per edge in the reference) is done once
//     per STAGED node while the tile is filled (3 div + 2 sqrt); values are identical because
//     the reference recomputes the very same per-node expressions for every incident edge.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <stdexcept>
#include <stdint.h>

#include "device_plan.hpp"

#ifndef MGCFD_KERNEL_NS
#error "compile with -DMGCFD_KERNEL_NS=exact|fast"
#endif

namespace mgcfd {
namespace MGCFD_KERNEL_NS {

// Diagnostic build only (tools/phase_half.py, -DMGCFD_PHASES): thread 0 of every workgroup adds the wall-clock ticks
// between phase boundaries of k_flux_half to its own slot.  The shipped build defines the marks away.
#ifdef MGCFD_PHASES
__device__ unsigned long long g_phase[4096 * 8];
__device__ unsigned long long g_phase_abs[4096 * 8];     // the LAST launch: [0] when every workgroup began, [1 + k] when it passed mark k (100 MHz ticks), [7] XCC_ID << 32 | HW_ID
#define PH_MARK(k) do { if (threadIdx.x == 0) { unsigned long long now_ = wall_clock64(); g_phase[(blockIdx.x & 4095) * 8 + (k)] += now_ - ph_last_; ph_last_ = now_; g_phase_abs[(blockIdx.x & 4095) * 8 + 1 + (k)] = now_; } } while (0)
#define PH_BEGIN() unsigned long long ph_last_ = wall_clock64(); if (threadIdx.x == 0) { g_phase[(blockIdx.x & 4095) * 8 + 7] += 1ull; g_phase_abs[(blockIdx.x & 4095) * 8] = ph_last_; \
        g_phase_abs[(blockIdx.x & 4095) * 8 + 7] = (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11))) << 32) | __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); }
#else
#define PH_MARK(k) do { } while (0)
#define PH_BEGIN() do { } while (0)
#endif

// Cache policy of the standalone flux launch's stores: write-through (`sc1`: the line leaves the XCD's L2 at once and is not
// kept).  Nothing in the launch reads `fluxes`, an XCD's L2 keeps nothing across a kernel boundary (profiles/r4_l2_across_launches.txt)
// and a launch otherwise ends with the write-back of everything it left dirty: 14.7 -> 14.1 us order-free, 16.1 -> 15.8 us
// bit-identical (profiles/r4_flux_levers.txt).  Experiment switches (tools/exp_flags.py builds libraries with -DMGCFD_EXP_*;
// the shipped build defines none of them).
#if defined(MGCFD_EXP_FLUX_ST_NT)
#define MGCFD_ST_FLUX(p, v) __builtin_nontemporal_store((v), (p))
#elif defined(MGCFD_EXP_FLUX_ST_PLAIN)
#define MGCFD_ST_FLUX(p, v) (*(p) = (v))
#else
#define MGCFD_ST_FLUX(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif
// ... and of every store of a node's new state (the fused stages, time_step, the transfers): write-through as well
// (sweep 55.0 -> 54.4 us, V-cycle 0.293 -> 0.287 ms; non-temporal stores: 57.1 us / 0.295 ms, profiles/r4_flux_levers.txt)
#if defined(MGCFD_EXP_STAGE_ST_PLAIN)
#define MGCFD_ST_STAGE(p, v) (*(p) = (v))
#elif defined(MGCFD_EXP_STAGE_ST_NT)
#define MGCFD_ST_STAGE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define MGCFD_ST_STAGE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif
#if defined(MGCFD_EXP_STATE_LD_NT)
#define MGCFD_LD_STATE(p) __builtin_nontemporal_load(p)
#else
#define MGCFD_LD_STATE(p) (*(p))
#endif

namespace {

constexpr double kGamma = 1.4;                 // src/Base/const.h:9
constexpr int kBlock = 256;

// ---- per-node derived state -------------------------------------------------------------
// cfd_loops.h:121-148.  Same expressions, same association as the reference.
struct Derived { double vx, vy, vz, p, speed, c, speed_sqd; };

__device__ __forceinline__ Derived derive(double rho, double mx, double my, double mz, double en)
{
    Derived d;
    d.vx = mx / rho;
    d.vy = my / rho;
    d.vz = mz / rho;
    d.speed_sqd = d.vx * d.vx + d.vy * d.vy + d.vz * d.vz;
    d.p = (kGamma - 1.0) * (en - 0.5 * rho * d.speed_sqd);
    d.speed = sqrt(d.speed_sqd);
    d.c = sqrt(kGamma * d.p / rho);
    return d;
}

// What the flux needs of one node: the 5 conserved variables plus what the reference derives
// from them for every incident edge.
struct NodeQ { double rho, mx, my, mz, en, vx, vy, vz, p, speed, c; };

__device__ __forceinline__ NodeQ make_nodeq(double rho, double mx, double my, double mz, double en)
{
    const Derived d = derive(rho, mx, my, mz, en);
    NodeQ r;
    r.rho = rho; r.mx = mx; r.my = my; r.mz = mz; r.en = en;
    r.vx = d.vx; r.vy = d.vy; r.vz = d.vz; r.p = d.p; r.speed = d.speed; r.c = d.c;
    return r;
}

__device__ __forceinline__ NodeQ load_and_derive(const double *__restrict__ q, int64_t stride, int64_t i)
{
    return make_nodeq(q[i], q[stride + i], q[2 * stride + i], q[3 * stride + i], q[4 * stride + i]);
}

__device__ __forceinline__ void store_conserved(double *__restrict__ q, int64_t stride, int64_t i, double rho,
                                                double mx, double my, double mz, double en)
{
    MGCFD_ST_STAGE(q + i, rho);
    MGCFD_ST_STAGE(q + stride + i, mx);
    MGCFD_ST_STAGE(q + 2 * stride + i, my);
    MGCFD_ST_STAGE(q + 3 * stride + i, mz);
    MGCFD_ST_STAGE(q + 4 * stride + i, en);
}

// LDS tile records: 12 doubles (96 B) per staged node, read back with 16-byte LDS loads.
constexpr int kLdsRecD2 = 6;               // double2 per record

// Within a record the three pairs of 16-byte quads are swapped when bit 3 of the slot is set: with a
// 24-dword record stride a fixed quad of slots s and s+8 would otherwise sit on the same 4 banks and
// a 16-lane ds_read_b128 group could reach only 8 of the 16 quad positions of the 256-B bank row
// (MI355X_MICROARCH.md, LDS); the swap makes all 16 reachable.
__device__ __forceinline__ void lds_store_record(double2 *tile, uint32_t slot, const NodeQ &n)
{
    double2 *rec = tile + slot * kLdsRecD2;
    const uint32_t b = (slot >> 3) & 1u;
    rec[0 ^ b] = make_double2(n.rho, n.mx);
    rec[1 ^ b] = make_double2(n.my, n.mz);
    rec[2 ^ b] = make_double2(n.en, n.vx);
    rec[3 ^ b] = make_double2(n.vy, n.vz);
    rec[4 ^ b] = make_double2(n.p, n.speed);
    rec[5 ^ b] = make_double2(n.c, 0.0);
}

__device__ __forceinline__ NodeQ lds_load_record(const double2 *tile, uint32_t slot)
{
    const uint32_t b = (slot >> 3) & 1u;
    const double2 *even = tile + slot * kLdsRecD2 + b;      // logical quads 0, 2, 4 (+0, +2, +4)
    const double2 *odd = tile + slot * kLdsRecD2 - b;       // logical quads 1, 3, 5 (+1, +3, +5)
    const double2 a = even[0], bq = odd[1], c = even[2], d = odd[3], e = even[4];
    const double f = odd[5].x;
    NodeQ r;
    r.rho = a.x; r.mx = a.y; r.my = bq.x; r.mz = bq.y; r.en = c.x; r.vx = c.y;
    r.vy = d.x; r.vz = d.y; r.p = e.x; r.speed = e.y; r.c = f;
    return r;
}

// The nine distinct flux-contribution components (cfd_loops.h:57-83); the momentum tensor is
// symmetric in storage: fmy.x = fmx.y, fmz.x = fmx.z, fmz.y = fmy.z.
struct FluxC { double xx, xy, xz, yy, yz, zz, ex, ey, ez; };

__device__ __forceinline__ FluxC flux_contribution(const NodeQ &q)
{
    FluxC f;
    f.xx = q.vx * q.mx + q.p;
    f.xy = q.vx * q.my;
    f.xz = q.vx * q.mz;
    f.yy = q.vy * q.my + q.p;
    f.yz = q.vy * q.mz;
    f.zz = q.vz * q.mz + q.p;
    const double de_p = q.en + q.p;
    f.ex = q.vx * de_p;
    f.ey = q.vy * de_p;
    f.ez = q.vz * de_p;
    return f;
}

// Minimum over the 64 lanes of a wave, in every lane; call it with all 64 lanes active (a disabled lane would
// be read as 0 by the DPP moves).  Cross-lane moves by DPP inside each row of 16
// lanes (no trip through the LDS crossbar as __shfl_xor takes, which would also queue behind the
// record stores of the tile kernels), then the four row minima are read as scalars.
template <int ctrl>
__device__ __forceinline__ double dpp_move(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp(static_cast<int>(b), ctrl, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(static_cast<int>(b >> 32), ctrl, 0xF, 0xF, true);
    return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo));
}

__device__ __forceinline__ double read_lane(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane(static_cast<int>(b), lane);
    const int hi = __builtin_amdgcn_readlane(static_cast<int>(b >> 32), lane);
    return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo));
}

__device__ __forceinline__ double wave_min(double v)
{
    v = fmin(v, dpp_move<0xB1>(v));      // quad_perm [1,0,3,2]: lane ^ 1
    v = fmin(v, dpp_move<0x4E>(v));      // quad_perm [2,3,0,1]: lane ^ 2  -> quads uniform
    v = fmin(v, dpp_move<0x141>(v));     // row_half_mirror: i <-> 7-i     -> groups of 8 uniform
    v = fmin(v, dpp_move<0x140>(v));     // row_mirror: i <-> 15-i         -> rows of 16 uniform
    return fmin(fmin(read_lane(v, 0), read_lane(v, 16)), fmin(read_lane(v, 32), read_lane(v, 48)));
}

__device__ __forceinline__ double wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// check_for_invalid_variables (validation.cpp:107-138) runs after every time_step and stops at the first bad cell
// in original order.  Here a bad cell leaves (launch sequence number, original id, code) in one word and the
// smallest wins: the earliest checked launch since the host last looked, then the smallest original id.
// `check` = that sequence number (>= 1, < 2^23; 0 switches the check off).
__device__ __forceinline__ unsigned long long err_key(int check, int32_t original_id, int code)
{
    return (static_cast<unsigned long long>(check) << 40) | (static_cast<unsigned long long>(uint32_t(original_id)) << 8) | unsigned(code);
}

// XCD-aware block order: hardware deals consecutive workgroups round-robin over the 8 XCDs
// (each with a private L2).  Give every XCD one CONTIGUOUS range of tiles instead, so that a
// tile's halo — owned by neighbouring tiles — is usually already in the same L2.  Pure speed:
// any placement is correct.
__device__ __forceinline__ unsigned xcd_contiguous_block(unsigned b, unsigned nb)
{
    const unsigned q = nb >> 3, r = nb & 7u, x = b & 7u, k = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

} // namespace

// ------------------------------------------------------------------------------------------
// initialize_variables (cfd_loops.h:44-55).  Runs over the padded length so the tail of every
// field holds valid numbers.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_init_variables(int64_t stride, FarField ff, double *__restrict__ q)
{
    const int64_t i = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    if (i >= stride) return;
    store_conserved(q, stride, i, ff.var[0], ff.var[1], ff.var[2], ff.var[3], ff.var[4]);
}

// ------------------------------------------------------------------------------------------
// compute_step_factor, first half (cfd_loops.cpp:98-125): sf = 0.5 * cbrt(vol) / (|v| + c) and
// the minimum over the level.  cbrt(vol) is static and precomputed on the host with the same
// libm the reference would call.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_step_factor_local(int64_t nel, int64_t stride, const double *__restrict__ q, const double *__restrict__ cbrt_vol,
                    double *__restrict__ step_factors, double *__restrict__ partial_min,
                    double *__restrict__ old_variables /* nullptr, or: fused copy<double>(old, variables) */)
{
    __shared__ double s_min[kBlock / 64];
    const int64_t i = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    double sf = __longlong_as_double(0x7FF0000000000000LL);          // +inf
    if (i < nel) {
        const double rho = q[i], mx = q[stride + i], my = q[2 * stride + i], mz = q[3 * stride + i], en = q[4 * stride + i];
        if (old_variables) {
            old_variables[i] = rho; old_variables[stride + i] = mx; old_variables[2 * stride + i] = my;
            old_variables[3 * stride + i] = mz; old_variables[4 * stride + i] = en;
        }
        const Derived d = derive(rho, mx, my, mz, en);
        const double dt = cbrt_vol[i] / (d.speed + d.c);
        sf = 0.5 * dt;
        step_factors[i] = sf;
    }
    // One partial minimum per workgroup; the consumer reduces them (min is order independent,
    // and one atomic word would serialise ~1200 updates).
    sf = wave_min(sf);
    if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = sf;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = s_min[0];
        for (int w = 1; w < kBlock / 64; w++) m = fmin(m, s_min[w]);
        partial_min[blockIdx.x] = m;
    }
}

// minimum of the partial minima, by every thread of a workgroup (result in all lanes)
__device__ __forceinline__ double block_min_of_partials(const double *__restrict__ partial_min, int n_partial)
{
    __shared__ double s_red[kBlock / 64];
    double m = __longlong_as_double(0x7FF0000000000000LL);
    for (int k = threadIdx.x; k < n_partial; k += kBlock) m = fmin(m, partial_min[k]);
    m = wave_min(m);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    double r = s_red[0];
    for (int w = 1; w < kBlock / 64; w++) r = fmin(r, s_red[w]);
    return r;
}

// One workgroup's minimum of v -> out[blockIdx.x]; every thread of the workgroup must call it.
__device__ __forceinline__ void block_min_to(double v, double *__restrict__ out)
{
    __shared__ double s_m[kBlock / 64];
    v = wave_min(v);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = s_m[0];
        for (int w = 1; w < kBlock / 64; w++) m = fmin(m, s_m[w]);
        out[blockIdx.x] = m;
    }
}

// first half of compute_step_factor for one node, as k_step_factor_local computes it
__device__ __forceinline__ double local_step_factor(double rho, double mx, double my, double mz, double en, double cbrt_vol)
{
    const Derived d = derive(rho, mx, my, mz, en);
    const double dt = cbrt_vol / (d.speed + d.c);
    return 0.5 * dt;
}

// partial minima -> one scalar (only needed where the scalar itself is the interface: the
// kernel-granular API and the multi-GPU all-reduce hook)
__global__ void __launch_bounds__(kBlock)
k_min_reduce(const double *__restrict__ partial_min, int n_partial, double *__restrict__ out)
{
    const double m = block_min_of_partials(partial_min, n_partial);
    if (threadIdx.x == 0) out[0] = m;
}

// second half (cfd_loops.cpp:146-156): step_factors[i] = min_dt / volumes[i]
__global__ void __launch_bounds__(kBlock)
k_step_factor_apply(int64_t nel, const double *__restrict__ min_dt_scalar,
                    const double *__restrict__ volumes, double *__restrict__ step_factors)
{
    const int64_t i = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    if (i >= nel) return;
    step_factors[i] = min_dt_scalar[0] / volumes[i];
}

// compute_step_factor_legacy (cfd_loops.cpp:37-61), mesh_name = fvcorr only
__global__ void __launch_bounds__(kBlock)
k_step_factor_legacy(int64_t nel, int64_t stride, const double *__restrict__ q, const double *__restrict__ volumes,
                     double *__restrict__ step_factors, double *__restrict__ old_variables /* nullptr or fused copy */)
{
    const int64_t i = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    if (i >= nel) return;
    const double rho = q[i], mx = q[stride + i], my = q[2 * stride + i], mz = q[3 * stride + i], en = q[4 * stride + i];
    if (old_variables) {
        old_variables[i] = rho; old_variables[stride + i] = mx; old_variables[2 * stride + i] = my;
        old_variables[3 * stride + i] = mz; old_variables[4 * stride + i] = en;
    }
    const Derived d = derive(rho, mx, my, mz, en);
    step_factors[i] = 0.5 / (sqrt(volumes[i]) * (d.speed + d.c));
}

// ------------------------------------------------------------------------------------------
// flux_tile: compute_flux_edge + compute_boundary_flux_edge + compute_wall_flux_edge
// (flux_loops.cpp:10-153) as one node-centred gather served from an LDS tile.
//
// One 256-thread workgroup = one tile = 256 consecutive nodes forming a compact cluster of
// the mesh graph (preprocess.cpp: cluster_order).  Phase 1: every staged node — the 256 own
// nodes (coalesced loads) and the tile's halo, the few hundred outside nodes its edges touch
// (gathered by id) — has its 5 conserved variables read from HBM ONCE, its velocity, pressure,
// |v| and speed of sound derived (cfd_loops.h:121-148) and the 11 values written to LDS as one
// 96-byte record.  Phase 2: every lane walks the incidence rows of its node (sliced ELLPACK:
// row r of a wave = the r-th incident edge of its 64 nodes, stored contiguously: a 16-bit
// tile-local neighbour slot, the 3 signed, halved edge-weight components and optionally the
// precomputed length factor = 26 or 34 B per entry),
// reads the neighbour's record from LDS and adds the edge flux.  Rows are in ORIGINAL edge
// order, internal edges first, then solid-wall faces, then far-field faces — exactly the order
// in which the reference's serial loops add into fluxes[node] — so the per-node sequential sum
// reproduces the reference's floating-point result bit for bit (exact build).
//
// `classes` selects the edge classes (bit0 internal, bit1 solid wall "-1", bit2 far field
// "-2"); ACC starts from the value already in `fluxes` (the reference's "+=" when the array is
// not known to be zero).  Halo nodes beyond the LDS capacity (ragged clusters
// only) are listed in a per-tile overflow table and read straight from HBM.
// ------------------------------------------------------------------------------------------
struct EdgeRow { uint32_t code; double fx, fy, fz, k; };

// Edge weights are stored [row][4 components][64 lanes]: fx, fy, fz (signed, halved) and
// k = -|e|*smoothing*0.5.  LOADK = false skips the k stream (8 of 34 bytes per entry) and
// recomputes it from fx,fy,fz — bit-identical either way, a bandwidth-for-arithmetic trade.
template <bool LOADK>
__device__ __forceinline__ EdgeRow load_row(const uint16_t *__restrict__ nbr, const double *__restrict__ w,
                                            int64_t row, int lane)
{
    EdgeRow e;
    const double *wr = w + (row << 8) + lane;
#ifdef MGCFD_EXP_ROW_LD_NT          // (experiment: the rows are read once — streamed past the L2 so that the state stays in it)
    e.code = __builtin_nontemporal_load(nbr + (row << 6) + lane);
    e.fx = __builtin_nontemporal_load(wr); e.fy = __builtin_nontemporal_load(wr + 64); e.fz = __builtin_nontemporal_load(wr + 128);
    e.k = LOADK ? __builtin_nontemporal_load(wr + 192) : 0.0;
#elif defined(MGCFD_EXP_ROW_LD_SC1)  // (experiment: agent-scope loads of the weights — past the XCD's L2, from the memory-side cache)
    e.code = nbr[(row << 6) + lane];
    e.fx = __hip_atomic_load(wr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    e.fy = __hip_atomic_load(wr + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    e.fz = __hip_atomic_load(wr + 128, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    e.k = LOADK ? __hip_atomic_load(wr + 192, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#else
    e.code = nbr[(row << 6) + lane];
    e.fx = wr[0]; e.fy = wr[64]; e.fz = wr[128];
    e.k = LOADK ? wr[192] : 0.0;
#endif
    return e;
}

__device__ __forceinline__ EdgeRow pad_row_entry()
{
    EdgeRow e;
    e.code = kT16Pad; e.fx = 0.0; e.fy = 0.0; e.fz = 0.0; e.k = 0.0;
    return e;
}

// Indexed weights (WMODE 2): a tile's internal edges are listed ONCE (preprocess.cpp: te_*; 24 bytes each, the a-side
// weights -0.5*e), and a row entry is 4 bytes: the neighbour's LDS slot (nbr16) and the position of the entry's edge in
// its tile's list (gat16, role bit = this node is the edge's b end: it sees the negated weights).  Both end points of
// an edge inside the tile read the same 24 bytes — the second one from L1/L2 — so a launch moves 24 B per edge and
// tile + 4 B per entry instead of 26-34 B per entry (two per edge); the length factor is recomputed from the weights.
struct RowRef { uint32_t code, pos; };

__device__ __forceinline__ RowRef load_ref(const uint16_t *__restrict__ nbr16, const uint16_t *__restrict__ gat16, int64_t row, int lane)
{
    RowRef r;
    r.code = nbr16[(row << 6) + lane];
    r.pos = gat16[(row << 6) + lane];
    return r;
}

__device__ __forceinline__ EdgeRow gather_weights(const double *__restrict__ tw, const RowRef &r)
{
    const uint32_t p = r.pos & kT16SlotMask;
    const bool pad = p == kT16Pad;
    const double *wp = tw + (pad ? 0u : p) * 3u;
    const double x = wp[0], y = wp[1], z = wp[2];
    const bool neg = (r.pos & kT16RoleB) != 0;
    EdgeRow e;
    e.code = r.code;
    // (padding must carry zero weights: its +-0.0 contribution is what leaves a sum that started at +0.0 unchanged)
    e.fx = pad ? 0.0 : (neg ? -x : x);
    e.fy = pad ? 0.0 : (neg ? -y : y);
    e.fz = pad ? 0.0 : (neg ? -z : z);
    e.k = 0.0;
    return e;
}

struct Flux5 { double d, mx, my, mz, en; };

// One edge's contribution to THIS node's flux (flux_kernel.elemfunc.c:130-189 seen from this
// node: "me" - "other"; the b-side sign is folded into fx,fy,fz by the plan, x - f*y == x + (-f)*y).
template <bool LOADK>
__device__ __forceinline__ Flux5 edge_flux(const NodeQ &me, const FluxC &fm, const NodeQ &ot, const EdgeRow &e)
{
    const FluxC fo = flux_contribution(ot);
    const bool me_is_b = (e.code & kT16RoleB) != 0;
    const double fx = e.fx, fy = e.fy, fz = e.fz;
    double k = e.k;
    if (!LOADK) {
        // The plan stores f = -+0.5*e, so sqrt(f.f) = |e|/2 exactly and -(|e|/2 * s) is the
        // reference's -|e|*s*0.5 bit for bit (power-of-two scalings commute with rounding).
        const double half_ewt = sqrt(fx * fx + fy * fy + fz * fz);           // flux_kernel.elemfunc.c:27
        k = -(half_ewt * double(0.2f));                                       // :130, smoothing_coefficient = double(0.2f)
    }
    // factor = k * (speed_a + speed_b + c_a + c_b), left-associated (:130-131); only the order
    // of the two sound speeds depends on which end this node is.
    const double c_a = me_is_b ? ot.c : me.c;
    const double c_b = me_is_b ? me.c : ot.c;
    const double factor = k * (((me.speed + ot.speed) + c_a) + c_b);
    Flux5 f;
    f.d = factor * (me.rho - ot.rho) + fx * (me.mx + ot.mx) + fy * (me.my + ot.my) + fz * (me.mz + ot.mz);
    f.en = factor * (me.en - ot.en) + fx * (fm.ex + fo.ex) + fy * (fm.ey + fo.ey) + fz * (fm.ez + fo.ez);
    f.mx = factor * (me.mx - ot.mx) + fx * (fm.xx + fo.xx) + fy * (fm.xy + fo.xy) + fz * (fm.xz + fo.xz);
    f.my = factor * (me.my - ot.my) + fx * (fm.xy + fo.xy) + fy * (fm.yy + fo.yy) + fz * (fm.yz + fo.yz);
    f.mz = factor * (me.mz - ot.mz) + fx * (fm.xz + fo.xz) + fy * (fm.yz + fo.yz) + fz * (fm.zz + fo.zz);
    return f;
}

// What a node does with its complete flux: store it, or (FUSE) apply time_step to it
// (cfd_loops.cpp:241-268) — same operations as k_time_step — and write the new state to fs.q_out.
template <bool FUSE>
__device__ __forceinline__ void finish_node(int64_t i, int64_t nel, int64_t stride, double a0, double a1, double a2,
                                            double a3, double a4, double *__restrict__ fluxes, const FusedStep &fs,
                                            double min_dt, unsigned t)
{
    if (!FUSE) {
        if (i < nel) {
#ifdef MGCFD_ABL_FREE_ST_AOS           /* diagnostic: the node's fluxes written as one 40-byte record (layout wrong) */
            *reinterpret_cast<double2 *>(fluxes + 5 * i) = make_double2(a0, a1);
            *reinterpret_cast<double2 *>(fluxes + 5 * i + 2) = make_double2(a2, a3);
            fluxes[5 * i + 4] = a4;
#else
            MGCFD_ST_FLUX(fluxes + i, a0); MGCFD_ST_FLUX(fluxes + stride + i, a1); MGCFD_ST_FLUX(fluxes + 2 * stride + i, a2);
            MGCFD_ST_FLUX(fluxes + 3 * stride + i, a3); MGCFD_ST_FLUX(fluxes + 4 * stride + i, a4);
#endif
        }
        return;
    }
    // ---- fused time_step: same operations as k_time_step on the flux just summed ----
    // (operands fetched here, not under the row loop: the loop already sits at the register budget
    //  of 3 waves per SIMD and hoisting these twelve registers makes it spill)
    double sf_next = __longlong_as_double(0x7FF0000000000000LL);          // +inf: lanes past nel
    double ss = 0.0;                                                      // this node's share of the residual sum of squares
    if (i < nel) {
        const double r0 = fs.old_variables[i], r1 = fs.old_variables[stride + i], r2 = fs.old_variables[2 * stride + i],
                     r3 = fs.old_variables[3 * stride + i], r4 = fs.old_variables[4 * stride + i];
        double sf;
        if (fs.partial_min) {                       // first stage: finish compute_step_factor (cfd_loops.cpp:137-156)
            sf = min_dt / fs.volumes[i];
            fs.step_factors[i] = sf;
        } else {
            sf = fs.step_factors[i];
        }
        const double factor = sf / fs.rk_div;
        const double rho = r0 + factor * a0, mx = r1 + factor * a1, my = r2 + factor * a2, mz = r3 + factor * a3,
                     en = r4 + factor * a4;
        // q_out may be the array old_variables points at (last stage, in place): this thread has read
        // its node's old values above and nobody else reads them
        store_conserved(fs.q_out, stride, i, rho, mx, my, mz, en);
        if (fs.residuals || fs.sumsq_partial) {
            const double d0 = rho - r0, d1 = mx - r1, d2 = my - r2, d3 = mz - r3, d4 = en - r4;
            if (fs.residuals) {                      // (null: the caller writes it on demand, solver.cpp settle_residuals)
                MGCFD_ST_STAGE(fs.residuals + i, d0); MGCFD_ST_STAGE(fs.residuals + stride + i, d1); MGCFD_ST_STAGE(fs.residuals + 2 * stride + i, d2);
                MGCFD_ST_STAGE(fs.residuals + 3 * stride + i, d3); MGCFD_ST_STAGE(fs.residuals + 4 * stride + i, d4);
            }
            if (fs.sumsq_partial) ss = (((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3) + d4 * d4;
        }
        if (fs.check) {
            const bool finite = isfinite(rho) && isfinite(mx) && isfinite(my) && isfinite(mz) && isfinite(en);
            int code = 0;
            if (!finite) code = 1;
            else if (rho < 0.0) code = 2;
            else if (en < 0.0) code = 3;
            if (code) atomicMin(fs.err, err_key(fs.check, fs.old_of_new[i], code));
        }
        // look-ahead: the next sweep's compute_step_factor starts from the state just produced
        if (fs.next_partial_min) {
            const Derived d = derive(rho, mx, my, mz, en);
            const double dt = fs.cbrt_vol[i] / (d.speed + d.c);          // k_step_factor_local
            sf_next = 0.5 * dt;
        } else if (fs.next_legacy_sf) {
            const Derived d = derive(rho, mx, my, mz, en);
            fs.next_legacy_sf[i] = 0.5 / (sqrt(fs.volumes[i]) * (d.speed + d.c));   // k_step_factor_legacy
        }
    }
    if (fs.next_partial_min || fs.sumsq_partial) {  // uniform: every thread of the workgroup takes part
        __shared__ double s_next[2][kBlock / 64];
        sf_next = wave_min(sf_next);
        ss = wave_sum(ss);
        if ((threadIdx.x & 63) == 0) { s_next[0][threadIdx.x >> 6] = sf_next; s_next[1][threadIdx.x >> 6] = ss; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double m = s_next[0][0], sum = s_next[1][0];
            for (int wv = 1; wv < kBlock / 64; wv++) { m = fmin(m, s_next[0][wv]); sum += s_next[1][wv]; }
            if (fs.next_partial_min) fs.next_partial_min[t] = m;
            if (fs.sumsq_partial) fs.sumsq_partial[t] = sum;
        }
    }
}

// The boundary rows of a node (after its internal rows): solid-wall faces, then far-field faces.
__device__ __forceinline__ void boundary_rows(const NodeQ &me, const FluxC &fm, const FarField &ff,
                                              const uint16_t *__restrict__ nbr16, const double *__restrict__ w,
                                              int64_t first_row, int32_t n_bnd, int lane, int classes, double &a0,
                                              double &a1, double &a2, double &a3, double &a4)
{
    for (int32_t r = 0; r < n_bnd; r++) {
        const EdgeRow e = load_row<false>(nbr16, w, first_row + r, lane);
        const double fx = e.fx, fy = e.fy, fz = e.fz;
        if (e.code == kT16Wall && (classes & 2)) {
            // flux_boundary_kernel.elemfunc.c:37-64: pressure force only
            a0 += 0.0;
            a1 += fx * me.p;
            a2 += fy * me.p;
            a3 += fz * me.p;
            a4 += 0.0;
        } else if (e.code == kT16Far && (classes & 4)) {
            // flux_wall_kernel.elemfunc.c:51-88: average with the far-field state
            a0 += fx * (ff.var[1] + me.mx) + fy * (ff.var[2] + me.my) + fz * (ff.var[3] + me.mz);
            a4 += fx * (ff.fc_de[0] + fm.ex) + fy * (ff.fc_de[1] + fm.ey) + fz * (ff.fc_de[2] + fm.ez);
            a1 += fx * (ff.fc_mx[0] + fm.xx) + fy * (ff.fc_mx[1] + fm.xy) + fz * (ff.fc_mx[2] + fm.xz);
            a2 += fx * (ff.fc_my[0] + fm.xy) + fy * (ff.fc_my[1] + fm.yy) + fz * (ff.fc_my[2] + fm.yz);
            a3 += fx * (ff.fc_mz[0] + fm.xz) + fy * (ff.fc_mz[1] + fm.yz) + fz * (ff.fc_mz[2] + fm.zz);
        }
    }
}

// ROLE (fused stages only): 0 = may finish compute_step_factor (first stage), 2 = may write the residual, its
// squares and the look-ahead (last stage), 1 = neither: the paths a stage cannot take are compiled out.
// TAIL: the level has long rows (TailPlan): the per-node loop stops at the tile's row limit and the workgroup
// evaluates the remaining entries together (see below).
// WMODE: 0 = the length factor k recomputed from the row's weights, 1 = k streamed with them, 2 = indexed weights (above).
// The workgroup's LDS: the staged records, the first stage's partial minima per wave, the last stage's per-wave sums
// (559 records x 96 B + 32 + 64 B = 53,760 B = 42 of the 1,280-byte granules LDS is handed out in: three workgroups per CU).
struct FluxTileLds {
    double2 tile[kTileCap * kLdsRecD2];
    double s_pm[kBlock / 64];
    double s_next[2][kBlock / 64];
};

// The kernel's body as a function of the workgroup's position in the launch (`block`): k_flux_tile calls it with blockIdx.x
// (tools/exp/sweep_flow.patch: a dataflow sweep calls it with the tile a workgroup has claimed).
template <int WMODE, bool FUSE, bool ACC, int ROLE, bool TAIL, bool PUSH>
__device__ __forceinline__ void
flux_tile_body(FluxTileLds &lds, const unsigned block,
               const double *__restrict__ q, const int32_t *__restrict__ tile_halo, uint32_t n_tiles, int32_t pad_row,
               int64_t stride, int64_t nel, const int32_t *__restrict__ slice_row0,
               const int32_t *__restrict__ rows_int, const int32_t *__restrict__ rows_bnd,
               const uint16_t *__restrict__ nbr16, const double *__restrict__ w,
               const int32_t *__restrict__ tile_ovf_ptr, const int32_t *__restrict__ tile_ovf, const FarField &ff,
               double *__restrict__ fluxes, int classes, const FusedStep &fs, const TailPlan &tp,
               const uint16_t *__restrict__ gat16, const int32_t *__restrict__ te_chunk_ptr, const double *__restrict__ te_w3,
               const StagePush &push /* PUSH instantiations only: the stage sends its own message (device_plan.hpp) */,
               const bool block_is_tile = false /* `block` IS the tile (tools/exp/sweep_flow.patch) */)
{
    constexpr bool LOADK = WMODE == 1;
    static_assert(!PUSH || FUSE, "only a fused stage sends its message");
    PH_BEGIN();
    constexpr bool IDXW = WMODE == 2;
    static_assert(!(IDXW && TAIL), "indexed weights: levels without long rows only");
    double2 *const tile = lds.tile;

    // FUSE: this launch is a whole Runge-Kutta stage — the node's complete flux never leaves
    // registers; time_step (cfd_loops.cpp:241-268) is applied to it at the end and the new state
    // goes to fs.q_out (a different buffer: neighbours still read the stage's input from q).
    double min_dt = 0.0;
    // First stage: the minimum over the step-factor partials.  Their loads go out with the prologue's (last, they
    // are not on its dependent chain), the reduction happens after the records are staged and shares the staging
    // barrier, so none of it adds to the prologue's latency chain.
    constexpr int kPartPre = 6;                                           // partials per thread held in registers (1,536 tiles)
    double *const s_pm = lds.s_pm;
    double pmv[kPartPre];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    unsigned t = block_is_tile ? block : xcd_contiguous_block(block, n_tiles);   // n_tiles == gridDim.x, without the hidden-argument load
    if (PUSH) {
        // boundary tiles in dispatch order first (their message leaves while the rest of the launch runs), the interior
        // tiles behind them in XCD-contiguous ranges
        const unsigned nb = unsigned(push.n_boundary);
        t = block < nb ? block : nb + xcd_contiguous_block(block - nb, n_tiles - nb);
    }
    if (FUSE && fs.tile_list) t = unsigned(fs.tile_list[t]);           // a launch over part of the level (uniform branch)
    const int64_t base = int64_t(t) * kTile;
    const int64_t i = base + tid;
    const int32_t slice = __builtin_amdgcn_readfirstlane(static_cast<int32_t>(i >> 6));

    // Issue order of the prologue's loads (they return in order, and a wait after a conditional load
    // is conservative, so none of them sits under a branch): the halo ids first — they head the only
    // dependent chain (ids -> halo state; fixed-stride table, the address needs only the tile number)
    // — then the own node's state, the first two rows' ids and weights, and the halo state by id.
    const int32_t *hrow = tile_halo + int64_t(t) * kHaloStride;
    const int32_t hid = hrow[tid];                                     // -1: no halo node for this thread
    const int32_t hid2 = (ROLE != 5 && tid < kHaloStride - kBlock) ? hrow[kBlock + tid] : -1;   // halo larger than the workgroup (rare)
    const int32_t row0 = slice_row0[slice];
    const int32_t n_int = (classes & 1) ? (TAIL ? tp.rows_main : rows_int)[slice] : 0;
    const int32_t n_bnd = rows_bnd[slice];
    const bool has_halo = hid >= 0;
    const int64_t hnode = has_halo ? int64_t(hid) : i;
    double o0, o1, o2, o3, o4, g0, g1, g2, g3, g4;
    EdgeRow e0, e1;
    // indexed weights: the first four rows' references go out right behind the halo ids (the weights they point at are
    // the second link of a chain as long as ids -> halo state); the tile's edge list starts at its first chunk
    RowRef i0{}, i1{}, i2{}, i3{};
    const double *tw = nullptr;
    if (IDXW) {
        i0 = load_ref(nbr16, gat16, n_int > 0 ? row0 : pad_row, lane);
        i1 = load_ref(nbr16, gat16, n_int > 1 ? row0 + 1 : pad_row, lane);
        i2 = load_ref(nbr16, gat16, n_int > 2 ? row0 + 2 : pad_row, lane);
        i3 = load_ref(nbr16, gat16, n_int > 3 ? row0 + 3 : pad_row, lane);
        tw = te_w3 + int64_t(te_chunk_ptr[t]) * (kEdgeChunk * 3);
    }
    if (FUSE && ROLE == 5) {
        // The input state of this stage does not exist in memory: it is the first stage's time_step,
        // old + (min_dt/volume/vin_div) * flux, applied here to every staged node (own and halo) from the
        // first stage's fluxes — the split sweep of a multi-rank run saves its separate time_step launch.
#pragma unroll
        for (int u = 0; u < kPartPre; u++) {
            const int k = threadIdx.x + u * kBlock;
            pmv[u] = fs.partial_min[k < fs.n_partial ? k : fs.n_partial - 1];
        }
        const double *od = fs.old_variables, *fl = fs.vin_flux;
        const double b0 = od[i], b1 = od[stride + i], b2 = od[2 * stride + i], b3 = od[3 * stride + i], b4 = od[4 * stride + i];
        const double f0 = fl[i], f1 = fl[stride + i], f2 = fl[2 * stride + i], f3 = fl[3 * stride + i], f4 = fl[4 * stride + i];
        const double vo = fs.volumes[i];
        const double c0 = od[hnode], c1 = od[stride + hnode], c2 = od[2 * stride + hnode], c3 = od[3 * stride + hnode],
                     c4 = od[4 * stride + hnode];
        const double h0 = fl[hnode], h1 = fl[stride + hnode], h2 = fl[2 * stride + hnode], h3 = fl[3 * stride + hnode],
                     h4 = fl[4 * stride + hnode];
        const double vh = fs.volumes[hnode];
        if (IDXW) { e0 = gather_weights(tw, i0); e1 = gather_weights(tw, i1); }
        else {
            e0 = load_row<LOADK>(nbr16, w, n_int > 0 ? row0 : pad_row, lane);
            e1 = load_row<LOADK>(nbr16, w, n_int > 1 ? row0 + 1 : pad_row, lane);
        }
        double pm = pmv[0];
#pragma unroll
        for (int u = 1; u < kPartPre; u++) pm = fmin(pm, pmv[u]);
        for (int k = threadIdx.x + kPartPre * kBlock; k < fs.n_partial; k += kBlock) pm = fmin(pm, fs.partial_min[k]);
        pm = wave_min(pm);
        if (lane == 0) s_pm[tid >> 6] = pm;
        __syncthreads();
        min_dt = s_pm[0];
        for (int wv = 1; wv < kBlock / 64; wv++) min_dt = fmin(min_dt, s_pm[wv]);
        const double fo = (min_dt / vo) / fs.vin_div, fh = (min_dt / vh) / fs.vin_div;      // k_time_step: factor = sf / rk_div
        o0 = b0 + fo * f0; o1 = b1 + fo * f1; o2 = b2 + fo * f2; o3 = b3 + fo * f3; o4 = b4 + fo * f4;
        g0 = c0 + fh * h0; g1 = c1 + fh * h1; g2 = c2 + fh * h2; g3 = c3 + fh * h3; g4 = c4 + fh * h4;
        if (fs.check_vin && i < nel) {                                  // the first stage's check_for_invalid_variables (its own, earlier, sequence number)
            const bool finite = isfinite(o0) && isfinite(o1) && isfinite(o2) && isfinite(o3) && isfinite(o4);
            int code = 0;
            if (!finite) code = 1;
            else if (o0 < 0.0) code = 2;
            else if (o4 < 0.0) code = 3;
            if (code) atomicMin(fs.err, err_key(fs.check_vin, fs.old_of_new[i], code));
        }
    } else {
        o0 = q[i]; o1 = q[stride + i]; o2 = q[2 * stride + i]; o3 = q[3 * stride + i]; o4 = q[4 * stride + i];
        g0 = q[hnode]; g1 = q[stride + hnode]; g2 = q[2 * stride + hnode]; g3 = q[3 * stride + hnode]; g4 = q[4 * stride + hnode];
        // (a row the slice does not have is read from pad_row, a row of padding after the last one: the
        //  load itself is never conditional, so the compiler can count the loads in flight exactly)
        if (IDXW) { e0 = gather_weights(tw, i0); e1 = gather_weights(tw, i1); }
        else {
            e0 = load_row<LOADK>(nbr16, w, n_int > 0 ? row0 : pad_row, lane);
            e1 = load_row<LOADK>(nbr16, w, n_int > 1 ? row0 + 1 : pad_row, lane);
        }
    }
    // (the step-factor partials: wanted only at the staging barrier, so requested after everything on the critical chain)
    if (FUSE && ROLE == 0) {                         // (role 0 is launched only with fs.partial_min set)
#pragma unroll
        for (int u = 0; u < kPartPre; u++) {
            const int k = threadIdx.x + u * kBlock;
            pmv[u] = fs.partial_min[k < fs.n_partial ? k : fs.n_partial - 1];     // clamped: a repeat does not change a minimum
        }
    }
    const NodeQ me = make_nodeq(o0, o1, o2, o3, o4);
    lds_store_record(tile, uint32_t(tid), me);
    // Unconditional: a thread without a halo node re-reads its own node and parks the copy in its
    // (unused) halo slot.  A branch here lets the compiler sink the gather loads below the own
    // record's derivation and serialise the two.
    lds_store_record(tile, uint32_t(kTile + tid), make_nodeq(g0, g1, g2, g3, g4));
    if (hid2 >= 0) lds_store_record(tile, uint32_t(kTile + kBlock + tid), load_and_derive(q, stride, hid2));

    const FluxC fm_pre = flux_contribution(me);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0;
    if (ACC) {
        a0 = fluxes[i]; a1 = fluxes[stride + i]; a2 = fluxes[2 * stride + i];
        a3 = fluxes[3 * stride + i]; a4 = fluxes[4 * stride + i];
    }
    const int32_t ovf0 = tile_ovf_ptr[t];
    if (FUSE && ROLE == 0) {                         // (role 0 is launched only with fs.partial_min set)
        double pm = pmv[0];
#pragma unroll
        for (int u = 1; u < kPartPre; u++) pm = fmin(pm, pmv[u]);
        for (int k = threadIdx.x + kPartPre * kBlock; k < fs.n_partial; k += kBlock) pm = fmin(pm, fs.partial_min[k]);
        pm = wave_min(pm);
        if (lane == 0) s_pm[tid >> 6] = pm;
    }
    PH_MARK(0);
    __syncthreads();
    PH_MARK(1);
    // (the four per-wave minima stay in LDS until the epilogue: nothing of them occupies a register across the row loop)

    // ---- phase 2: incidence rows two at a time (independent arithmetic, ordered accumulation),
    //      ids/weights fetched two rows ahead ----
    // The row pair's work (rows e0, e1 -> a0..a4, strictly in row order = the reference's summation order).
    // ELL padding carries zero weights, so on a sum that started at +0.0 its +-0.0 contribution changes no
    // bit (x + y = -0.0 only if both are); a sum read from memory (ACC) may be -0.0 and skips padding.
    // With long rows (TAIL) an entry the plan moved to the workgroup's list is blanked in nbr16 only (its weights stay,
    // other kernels read them), so there too a padding code skips the add.
    // Halo nodes beyond the LDS tile (ragged clusters) are read from HBM.
#define MGCFD_ROW_PAIR()                                                                                     \
    do {                                                                                                     \
            const uint32_t s0 = e0.code & kT16SlotMask, s1 = e1.code & kT16SlotMask;                         \
            const bool v0 = s0 != kT16Pad, v1 = s1 != kT16Pad;                                               \
            const bool o0 = v0 && s0 >= uint32_t(kTileCap), o1 = v1 && s1 >= uint32_t(kTileCap);             \
            NodeQ n0, n1;                                                                                    \
            if (!TAIL && __builtin_expect(__any(o0 || o1), 0)) {      /* (long rows: such entries are on the list) */ \
                n0 = o0 ? load_and_derive(q, stride, tile_ovf[ovf0 + int32_t(s0) - kTileCap])                \
                        : lds_load_record(tile, v0 ? s0 : uint32_t(tid));                                    \
                n1 = o1 ? load_and_derive(q, stride, tile_ovf[ovf0 + int32_t(s1) - kTileCap])                \
                        : lds_load_record(tile, v1 ? s1 : uint32_t(tid));                                    \
            } else {                                                                                         \
                n0 = lds_load_record(tile, v0 ? s0 : uint32_t(tid));                                         \
                n1 = lds_load_record(tile, v1 ? s1 : uint32_t(tid));                                         \
            }                                                                                                \
            Flux5 f0, f1;                                                                                    \
            f0 = edge_flux<LOADK>(me, fm_pre, n0, e0);                                                       \
            f1 = edge_flux<LOADK>(me, fm_pre, n1, e1);                                                       \
            if (ACC || TAIL) {                                                                               \
                a0 = v0 ? a0 + f0.d : a0;   a1 = v0 ? a1 + f0.mx : a1;   a2 = v0 ? a2 + f0.my : a2;          \
                a3 = v0 ? a3 + f0.mz : a3;  a4 = v0 ? a4 + f0.en : a4;                                       \
                a0 = v1 ? a0 + f1.d : a0;   a1 = v1 ? a1 + f1.mx : a1;   a2 = v1 ? a2 + f1.my : a2;          \
                a3 = v1 ? a3 + f1.mz : a3;  a4 = v1 ? a4 + f1.en : a4;                                       \
            } else {                                                                                         \
                a0 += f0.d; a1 += f0.mx; a2 += f0.my; a3 += f0.mz; a4 += f0.en;                              \
                a0 += f1.d; a1 += f1.mx; a2 += f1.my; a3 += f1.mz; a4 += f1.en;                              \
            }                                                                                                \
    } while (0)
    // long rows: what the workgroup's list pass needs first is requested here and arrives while the rows are walked
    int32_t tl_b = 0, tl_e = 0, tl_nb = 0, tl_nc = 0;
    double2 tl_r0 = make_double2(0.0, 0.0), tl_r1 = tl_r0, tl_r2 = tl_r0;
    if (TAIL && (classes & 1)) {
        tl_b = tp.tile_ptr[t]; tl_e = tp.tile_ptr[t + 1];
        tl_nb = tp.begin[i]; tl_nc = tp.count[i];
        const int32_t e_first = tl_b + tid;
        if (e_first < tl_e) { tl_r0 = tp.rec[3 * int64_t(e_first)]; tl_r1 = tp.rec[3 * int64_t(e_first) + 1]; tl_r2 = tp.rec[3 * int64_t(e_first) + 2]; }
    }
    double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0, r4 = 0.0, sfv = 0.0;       // fused stages: the time_step operands
    {
        // every pair but the last, each prefetching the pair after it
        int32_t r = 0;
        if (IDXW) {
            // references two pairs ahead, the weights they point at one pair ahead
            for (; r + 2 < n_int; r += 2) {
                const RowRef i4 = load_ref(nbr16, gat16, r + 4 < n_int ? row0 + r + 4 : pad_row, lane);
                const RowRef i5 = load_ref(nbr16, gat16, r + 5 < n_int ? row0 + r + 5 : pad_row, lane);
                const EdgeRow e2 = gather_weights(tw, i2), e3 = gather_weights(tw, i3);
                MGCFD_ROW_PAIR();
                e0 = e2; e1 = e3; i2 = i4; i3 = i5;
            }
        } else
        for (; r + 2 < n_int; r += 2) {
            const EdgeRow e2 = load_row<LOADK>(nbr16, w, row0 + r + 2, lane);
            const EdgeRow e3 = load_row<LOADK>(nbr16, w, r + 3 < n_int ? row0 + r + 3 : pad_row, lane);
            MGCFD_ROW_PAIR();
            e0 = e2; e1 = e3;
        }
        // The last pair has nothing left to prefetch: in the fused stages the registers its prefetch would
        // have used take the time_step operands instead, which arrive while the pair is being summed.
        if (FUSE) {
            if (ROLE == 0) {
                // the first stage's input IS the sweep's start state (the launcher checks q == old_variables): the node's own
                // record, live in registers for every edge anyway, holds the five values
                r0 = me.rho; r1 = me.mx; r2 = me.my; r3 = me.mz; r4 = me.en;
            } else {
                r0 = fs.old_variables[i]; r1 = fs.old_variables[stride + i]; r2 = fs.old_variables[2 * stride + i];
                r3 = fs.old_variables[3 * stride + i]; r4 = fs.old_variables[4 * stride + i];
            }
            sfv = ((ROLE == 0 || ROLE == 5) ? fs.volumes : fs.step_factors)[i];
        }
        if (r < n_int) MGCFD_ROW_PAIR();
    }
#undef MGCFD_ROW_PAIR
    PH_MARK(2);

    if (TAIL && (classes & 1) && tl_e > tl_b) {                            // (uniform)
        // Long rows.  The entries beyond the tile's row limit, one per thread whatever node they belong to: the same
        // edge_flux from the same operands (the owner's record is read back from LDS); then every owner adds its own
        // entries in row order — the accumulation order is untouched, only the evaluation is spread over the
        // workgroup instead of waiting for the highest-degree lanes.  The results of the last round of 256 entries
        // (usually the only one) are handed over through LDS, where the records are dead by then; earlier rounds
        // go through a global scratch.
        const int32_t last0 = tl_b + ((tl_e - tl_b - 1) / kBlock) * kBlock;   // first entry of the last round
        int32_t e = tl_b + tid;
        double2 r0 = tl_r0, r1 = tl_r1, r2 = tl_r2;
        Flux5 fl;
        fl.d = 0.0; fl.mx = 0.0; fl.my = 0.0; fl.mz = 0.0; fl.en = 0.0;
        while (e < tl_e) {
            const double2 c0 = r0, c1 = r1, c2 = r2;
            const int32_t en = e + kBlock;
            if (en < tl_e) { r0 = tp.rec[3 * int64_t(en)]; r1 = tp.rec[3 * int64_t(en) + 1]; r2 = tp.rec[3 * int64_t(en) + 2]; }   // next round's entry
            const uint32_t word = static_cast<uint32_t>(__double_as_longlong(c2.x));
            const uint32_t own = word & 0xFFFFu;
            if (own != kT16Pad) {
                EdgeRow er;
                er.code = word >> 16;
                er.fx = c0.x; er.fy = c0.y; er.fz = c1.x; er.k = LOADK ? c1.y : 0.0;
                const uint32_t s = er.code & kT16SlotMask;
                const NodeQ mo = lds_load_record(tile, own);
                const NodeQ ot = s >= uint32_t(kTileCap) ? load_and_derive(q, stride, tile_ovf[ovf0 + int32_t(s) - kTileCap])
                                                         : lds_load_record(tile, s);
                const Flux5 f = edge_flux<LOADK>(mo, flux_contribution(mo), ot, er);
                if (e >= last0) {
                    fl = f;
                } else {
                    double2 *out = tp.flux + 3 * int64_t(e);
                    out[0] = make_double2(f.d, f.mx); out[1] = make_double2(f.my, f.mz); out[2] = make_double2(f.en, 0.0);
                }
            }
            e = en;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's scratch stores have left
        __syncthreads();                                                  // ... and nobody reads a record any more
        tile[3 * tid] = make_double2(fl.d, fl.mx); tile[3 * tid + 1] = make_double2(fl.my, fl.mz); tile[3 * tid + 2] = make_double2(fl.en, 0.0);
        __syncthreads();
        // (a tile's scratch range is whole 128-byte lines that only this workgroup touches, and it reads them
        //  only now: no stale copy can sit in this CU's L1)
        // ordered adds: first the node's entries of the earlier rounds (global), then those of the last round (LDS)
        const int32_t n_glob = min(tl_nc, max(0, last0 - tl_nb));
#define MGCFD_ADD_ENTRIES(in, count)                                                                              \
        do {                                                                                                      \
            int32_t k_ = 0;                                                                                       \
            for (; k_ + 3 <= (count); k_ += 3) {           /* the records fetched three entries ahead of the sums */ \
                const double2 u0 = in[0], u1 = in[1], u2 = in[2], v0 = in[3], v1 = in[4], v2 = in[5],              \
                              w0 = in[6], w1 = in[7], w2 = in[8];                                                 \
                in += 9;                                                                                          \
                a0 += u0.x; a1 += u0.y; a2 += u1.x; a3 += u1.y; a4 += u2.x;                                       \
                a0 += v0.x; a1 += v0.y; a2 += v1.x; a3 += v1.y; a4 += v2.x;                                       \
                a0 += w0.x; a1 += w0.y; a2 += w1.x; a3 += w1.y; a4 += w2.x;                                       \
            }                                                                                                     \
            for (; k_ < (count); k_++) {                                                                          \
                const double2 u0 = in[0], u1 = in[1], u2 = in[2];                                                 \
                in += 3;                                                                                          \
                a0 += u0.x; a1 += u0.y; a2 += u1.x; a3 += u1.y; a4 += u2.x;                                       \
            }                                                                                                     \
        } while (0)
        if (n_glob > 0) {
            const double2 *in = tp.flux + 3 * int64_t(tl_nb);
            MGCFD_ADD_ENTRIES(in, n_glob);
        }
        {
            const double2 *in = tile + 3 * (tl_nb + n_glob - last0);
            const int32_t n_lds = tl_nc - n_glob;
            MGCFD_ADD_ENTRIES(in, n_lds);
        }
#undef MGCFD_ADD_ENTRIES
    }

    if ((classes & 6) && n_bnd > 0) {
        // The reference runs ALL solid-wall faces, then ALL far-field faces; the plan lists a
        // node's faces in exactly that order, so one walk over the rows keeps the per-node order.
        const int32_t first_bnd = rows_int[slice];
        const FluxC fm = fm_pre;
        for (int32_t r = 0; r < n_bnd; r++) {
            const EdgeRow e = load_row<false>(nbr16, w, int64_t(row0) + first_bnd + r, lane);
            const double fx = e.fx, fy = e.fy, fz = e.fz;
            if (e.code == kT16Wall && (classes & 2)) {
                // flux_boundary_kernel.elemfunc.c:37-64: pressure force only
                a0 += 0.0;
                a1 += fx * me.p;
                a2 += fy * me.p;
                a3 += fz * me.p;
                a4 += 0.0;
            } else if (e.code == kT16Far && (classes & 4)) {
                // flux_wall_kernel.elemfunc.c:51-88: average with the far-field state
                a0 += fx * (ff.var[1] + me.mx) + fy * (ff.var[2] + me.my) + fz * (ff.var[3] + me.mz);
                a4 += fx * (ff.fc_de[0] + fm.ex) + fy * (ff.fc_de[1] + fm.ey) + fz * (ff.fc_de[2] + fm.ez);
                a1 += fx * (ff.fc_mx[0] + fm.xx) + fy * (ff.fc_mx[1] + fm.xy) + fz * (ff.fc_mx[2] + fm.xz);
                a2 += fx * (ff.fc_my[0] + fm.xy) + fy * (ff.fc_my[1] + fm.yy) + fz * (ff.fc_my[2] + fm.yz);
                a3 += fx * (ff.fc_mz[0] + fm.xz) + fy * (ff.fc_mz[1] + fm.yz) + fz * (ff.fc_mz[2] + fm.zz);
            }
        }
    }

    if (!FUSE) {
        if (i < nel) {
            MGCFD_ST_FLUX(fluxes + i, a0); MGCFD_ST_FLUX(fluxes + stride + i, a1); MGCFD_ST_FLUX(fluxes + 2 * stride + i, a2);
            MGCFD_ST_FLUX(fluxes + 3 * stride + i, a3); MGCFD_ST_FLUX(fluxes + 4 * stride + i, a4);
        }
        PH_MARK(3);
        return;
    }
    // ---- fused time_step: same operations as k_time_step on the flux just summed ----
    double sf_next = __longlong_as_double(0x7FF0000000000000LL);          // +inf: lanes past nel
    double ss = 0.0;                                                      // this node's share of the residual sum of squares
    if (i < nel) {
        double sf = sfv;                            // (operands: requested before the last row pair)
        if (ROLE == 0 || ROLE == 5) {               // first stage (or the stage that absorbed it): finish compute_step_factor (cfd_loops.cpp:137-156)
            min_dt = s_pm[0];
            for (int wv = 1; wv < kBlock / 64; wv++) min_dt = fmin(min_dt, s_pm[wv]);
            sf = min_dt / sfv;                      // sfv holds the volume
            fs.step_factors[i] = sf;
        }
        const double factor = sf / fs.rk_div;
        const double rho = r0 + factor * a0, mx = r1 + factor * a1, my = r2 + factor * a2, mz = r3 + factor * a3,
                     en = r4 + factor * a4;
        // q_out may be the array old_variables points at (last stage, in place): this thread has read
        // its node's old values above and nobody else reads them
        store_conserved(fs.q_out, stride, i, rho, mx, my, mz, en);
        if (ROLE >= 2 && ROLE <= 4) {               // last stage: residual (validation.cpp:77-89)
            const double d0 = rho - r0, d1 = mx - r1, d2 = my - r2, d3 = mz - r3, d4 = en - r4;
            if (fs.residuals) {                      // (null: the caller writes it on demand, solver.cpp settle_residuals)
                MGCFD_ST_STAGE(fs.residuals + i, d0); MGCFD_ST_STAGE(fs.residuals + stride + i, d1); MGCFD_ST_STAGE(fs.residuals + 2 * stride + i, d2);
                MGCFD_ST_STAGE(fs.residuals + 3 * stride + i, d3); MGCFD_ST_STAGE(fs.residuals + 4 * stride + i, d4);
            }
            if (fs.sumsq_partial) ss = (((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3) + d4 * d4;
        }
        if (fs.check) {
            const bool finite = isfinite(rho) && isfinite(mx) && isfinite(my) && isfinite(mz) && isfinite(en);
            int code = 0;
            if (!finite) code = 1;
            else if (rho < 0.0) code = 2;
            else if (en < 0.0) code = 3;
            if (code) atomicMin(fs.err, err_key(fs.check, fs.old_of_new[i], code));
        }
        // look-ahead: the next sweep's compute_step_factor starts from the state just produced
        if (ROLE == 3) {
            const Derived d = derive(rho, mx, my, mz, en);
            const double dt = fs.cbrt_vol[i] / (d.speed + d.c);          // k_step_factor_local
            sf_next = 0.5 * dt;
        } else if (ROLE == 4) {
            const Derived d = derive(rho, mx, my, mz, en);
            fs.next_legacy_sf[i] = 0.5 / (sqrt(fs.volumes[i]) * (d.speed + d.c));   // k_step_factor_legacy
        }
        if (PUSH && block < unsigned(push.n_boundary)) {
            // this node into the ghost slots of the neighbours that hold it
            for (int32_t e = push.send_ptr[i]; e < push.send_ptr[i + 1]; e++) {
                const int k = push.send_peer[e];
                const int64_t g = push.send_target[e];
                double *dst = push.peers.base[0];
                int64_t ps = push.peers.stride[0];
#pragma unroll
                for (int p = 1; p < kMaxPushPeers; p++) if (p == k) { dst = push.peers.base[p]; ps = push.peers.stride[p]; }
                dst[g] = rho; dst[ps + g] = mx; dst[2 * ps + g] = my; dst[3 * ps + g] = mz; dst[4 * ps + g] = en;
            }
        }
    }
    if (PUSH && block < unsigned(push.n_boundary)) {              // (uniform per workgroup)
        // as k_halo_push_flags: the stores acknowledged, the workgroup counted off, the last boundary tile raises the flags
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence_system();
            const unsigned done = __hip_atomic_fetch_add(push.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (done == unsigned(push.n_boundary) - 1u) {
                __hip_atomic_store(push.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __threadfence_system();
#pragma unroll
                for (int p = 0; p < kMaxPushPeers; p++)
                    if (p < push.flags.n) __hip_atomic_store(push.flags.flag[p], push.flags.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    if (ROLE == 3 || (ROLE >= 2 && ROLE <= 4 && fs.sumsq_partial)) {  // uniform: every thread of the workgroup takes part
        double (*const s_next)[kBlock / 64] = lds.s_next;
        if (ROLE == 3) sf_next = wave_min(sf_next);
        if (fs.sumsq_partial) ss = wave_sum(ss);
        if ((threadIdx.x & 63) == 0) { s_next[0][threadIdx.x >> 6] = sf_next; s_next[1][threadIdx.x >> 6] = ss; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double m = s_next[0][0], sum = s_next[1][0];
            for (int wv = 1; wv < kBlock / 64; wv++) { m = fmin(m, s_next[0][wv]); sum += s_next[1][wv]; }
            if (ROLE == 3) fs.next_partial_min[t] = m;
            if (fs.sumsq_partial) fs.sumsq_partial[t] = sum;
        }
    }
    PH_MARK(3);
}

template <int MINW, int WMODE, bool FUSE, bool ACC, int ROLE, bool TAIL, bool PUSH = false>
__global__ void __launch_bounds__(kBlock, MINW)
k_flux_tile(// the first 16 dwords of the arguments are preloaded into SGPRs at wave launch (Makefile:
            // -amdgpu-kernarg-preload-count): what the first loads of the prologue need comes first
            const double *__restrict__ q, const int32_t *__restrict__ tile_halo, uint32_t n_tiles, int32_t pad_row,
            int64_t stride, int64_t nel, const int32_t *__restrict__ slice_row0,
            const int32_t *__restrict__ rows_int, const int32_t *__restrict__ rows_bnd,
            const uint16_t *__restrict__ nbr16, const double *__restrict__ w,
            const int32_t *__restrict__ tile_ovf_ptr, const int32_t *__restrict__ tile_ovf, FarField ff,
            double *__restrict__ fluxes, int classes, FusedStep fs, TailPlan tp,
            const uint16_t *__restrict__ gat16, const int32_t *__restrict__ te_chunk_ptr, const double *__restrict__ te_w3,
            StagePush push)
{
    __shared__ FluxTileLds lds;
    flux_tile_body<WMODE, FUSE, ACC, ROLE, TAIL, PUSH>(lds, blockIdx.x, q, tile_halo, n_tiles, pad_row, stride, nel, slice_row0, rows_int, rows_bnd, nbr16, w,
                                                       tile_ovf_ptr, tile_ovf, ff, fluxes, classes, fs, tp, gat16, te_chunk_ptr, te_w3, push);
}

// ------------------------------------------------------------------------------------------
// flux_edge_once: the same three loops with every internal edge evaluated ONCE per tile.
//
// k_flux_tile reads an edge's weights and evaluates its flux twice, once from each end.  Here a
// tile lists the internal edges that touch it once (preprocess.cpp, te_*), in ORIGINAL edge
// order; with the node records staged as before, thread p % 256 evaluates edge p of the list
// (a-side form, flux_kernel.elemfunc.c:130-161) and keeps the five results in registers.  When
// all edges are done the records are dead: the fluxes replace them in LDS (40 B each, position
// p), and every node then sums its incident edges from LDS in its own row order = the reference's
// accumulation order; the b end adds -F, which is what the reference's b-side expressions
// (:170-189) evaluate to bit for bit (every term negates exactly).  An edge cut by a tile
// boundary is evaluated by both tiles from identical operands.  Per tile this moves ~36 B per
// listed edge + 2 B per row entry instead of 34 B per row entry (two per edge), and does ~0.6 of
// the flux arithmetic.
// ------------------------------------------------------------------------------------------
struct TileEdge { uint32_t sa, sb; double fx, fy, fz, k; };
constexpr int kGatherPre = 8;             // gather-list rows held in registers from the start of the kernel

template <bool LOADK>
__device__ __forceinline__ TileEdge load_tile_edge(const uint16_t *__restrict__ te_slots, const double *__restrict__ te_w,
                                                   int64_t chunk, int tid)
{
    TileEdge e;
    e.sa = te_slots[(chunk * 2) * kEdgeChunk + tid];
    e.sb = te_slots[(chunk * 2 + 1) * kEdgeChunk + tid];
    const double *wr = te_w + chunk * (4 * kEdgeChunk) + tid;
    e.fx = wr[0]; e.fy = wr[kEdgeChunk]; e.fz = wr[2 * kEdgeChunk];
    e.k = LOADK ? wr[3 * kEdgeChunk] : 0.0;
    return e;
}

__device__ __forceinline__ TileEdge no_tile_edge()
{
    TileEdge e;
    e.sa = kT16Pad; e.sb = kT16Pad; e.fx = 0.0; e.fy = 0.0; e.fz = 0.0; e.k = 0.0;
    return e;
}

template <bool LOADK, bool FUSE, bool ACC>
__global__ void __launch_bounds__(kBlock, 3)
k_flux_edge_once(const double *__restrict__ q, const int32_t *__restrict__ tile_halo, uint32_t n_tiles, int32_t pad_row,
                 int64_t stride, int64_t nel, const int32_t *__restrict__ te_chunk_ptr,
                 const int32_t *__restrict__ te_count, int32_t pad_chunk, int classes,
                 const int32_t *__restrict__ slice_row0,
                 const int32_t *__restrict__ rows_int, const int32_t *__restrict__ rows_bnd,
                 const uint16_t *__restrict__ nbr16, const double *__restrict__ w, const uint16_t *__restrict__ gat16,
                 const uint16_t *__restrict__ te_slots, const double *__restrict__ te_w,
                 const int32_t *__restrict__ tile_ovf_ptr, const int32_t *__restrict__ tile_ovf, FarField ff,
                 double *__restrict__ fluxes, FusedStep fs)
{
    __shared__ double2 tile[kTileCap * kLdsRecD2];

    double min_dt = 0.0;
    if (FUSE && fs.partial_min) min_dt = block_min_of_partials(fs.partial_min, fs.n_partial);

    const unsigned t = xcd_contiguous_block(blockIdx.x, n_tiles);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_base = tid & ~63;                 // first list position of this wave within a chunk
    const int64_t base = int64_t(t) * kTile;
    const int64_t i = base + tid;
    const int32_t slice = __builtin_amdgcn_readfirstlane(static_cast<int32_t>(i >> 6));

    // load issue order as in k_flux_tile: halo ids, own state, then the bulk prefetches (first two
    // chunks of the edge list, the node's gather list), then the halo state by id; nothing under a branch
    const int32_t *hrow = tile_halo + int64_t(t) * kHaloStride;
    const int32_t hid = hrow[tid];                                     // -1: no halo node for this thread
    const int32_t hid2 = tid < kHaloStride - kBlock ? hrow[kBlock + tid] : -1;   // halo larger than the workgroup (rare)
    const double o0 = q[i], o1 = q[stride + i], o2 = q[2 * stride + i], o3 = q[3 * stride + i], o4 = q[4 * stride + i];
    const int32_t n_te = te_count[t];
    const int64_t chunk0 = te_chunk_ptr[t];
    // (a chunk this wave has no edges in is read from pad_chunk, a chunk of padding after the last
    //  one: the loads are never conditional, so the compiler can count the loads in flight exactly)
    TileEdge e_cur = load_tile_edge<LOADK>(te_slots, te_w, wave_base < n_te ? chunk0 : pad_chunk, tid);
    TileEdge e_nxt = load_tile_edge<LOADK>(te_slots, te_w, kEdgeChunk + wave_base < n_te ? chunk0 + 1 : pad_chunk, tid);

    // the node's gather list (positions of its incident edges in the tile's list): the first rows
    // are fetched now, long before phase 4 needs them
    const int32_t row0 = slice_row0[slice];
    const int32_t n_int = rows_int[slice];
    const int32_t n_bnd = rows_bnd[slice];
    const uint16_t *grow = gat16 + (int64_t(row0) << 6) + lane;
    uint32_t gc[kGatherPre];
#pragma unroll
    for (int k = 0; k < kGatherPre; k++) gc[k] = gat16[(int64_t(k < n_int ? row0 + k : pad_row) << 6) + lane];

    // ---- phase 1: stage + derive, as k_flux_tile ----
    const bool has_halo = hid >= 0;
    const int64_t hnode = has_halo ? int64_t(hid) : i;
    {
        const double g0 = q[hnode], g1 = q[stride + hnode], g2 = q[2 * stride + hnode], g3 = q[3 * stride + hnode],
                     g4 = q[4 * stride + hnode];
        lds_store_record(tile, uint32_t(tid), make_nodeq(o0, o1, o2, o3, o4));
        lds_store_record(tile, uint32_t(kTile + tid), make_nodeq(g0, g1, g2, g3, g4));   // unconditional, see k_flux_tile
    }
    if (hid2 >= 0) lds_store_record(tile, uint32_t(kTile + kBlock + tid), load_and_derive(q, stride, hid2));
    const int32_t ovf0 = tile_ovf_ptr[t];
    __syncthreads();

    // ---- phase 2: one edge per thread and chunk ----
    Flux5 F[kMaxEdgeChunks];
#pragma unroll
    for (int c = 0; c < kMaxEdgeChunks; c++) {
        TileEdge e_n2 = no_tile_edge();
        if (c + 2 < kMaxEdgeChunks)                                    // compile-time
            e_n2 = load_tile_edge<LOADK>(te_slots, te_w, (c + 2) * kEdgeChunk + wave_base < n_te ? chunk0 + c + 2 : pad_chunk, tid);
        F[c].d = 0.0; F[c].mx = 0.0; F[c].my = 0.0; F[c].mz = 0.0; F[c].en = 0.0;
        if (c * kEdgeChunk + wave_base < n_te) {                     // wave-uniform
            const bool v = e_cur.sa != kT16Pad;                      // the list's last chunk is padded
            const uint32_t sa = v ? e_cur.sa : uint32_t(tid), sb = v ? e_cur.sb : uint32_t(tid);
            const bool oa = sa >= uint32_t(kTileCap), ob = sb >= uint32_t(kTileCap);
            NodeQ A, B;
            if (__builtin_expect(__any(oa || ob), 0)) {
                // ragged cluster: an end point did not fit the LDS tile, read it from HBM
                A = oa ? load_and_derive(q, stride, tile_ovf[ovf0 + int32_t(sa) - kTileCap]) : lds_load_record(tile, sa);
                B = ob ? load_and_derive(q, stride, tile_ovf[ovf0 + int32_t(sb) - kTileCap]) : lds_load_record(tile, sb);
            } else {
                A = lds_load_record(tile, sa);
                B = lds_load_record(tile, sb);
            }
            EdgeRow er;
            er.code = 0;                                             // role a
            er.fx = e_cur.fx; er.fy = e_cur.fy; er.fz = e_cur.fz; er.k = e_cur.k;
            F[c] = edge_flux<LOADK>(A, flux_contribution(A), B, er);
        }
        e_cur = e_nxt; e_nxt = e_n2;
    }

    // ---- phase 3: the records are dead; the edge fluxes take their place ----
    double *fb = reinterpret_cast<double *>(tile);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kMaxEdgeChunks; c++) {
        if (c * kEdgeChunk + wave_base < n_te) {
            double *slot = fb + (c * kEdgeChunk + tid) * 5;
            slot[0] = F[c].d; slot[1] = F[c].mx; slot[2] = F[c].my; slot[3] = F[c].mz; slot[4] = F[c].en;
        }
    }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0;
    if (ACC) {
        a0 = fluxes[i]; a1 = fluxes[stride + i]; a2 = fluxes[2 * stride + i];
        a3 = fluxes[3 * stride + i]; a4 = fluxes[4 * stride + i];
    }
    __syncthreads();

    // ---- phase 4: every node sums its incident edges in row order ----
    auto add_entry = [&](uint32_t code) {
        const uint32_t p = code & kT16SlotMask;
        if (p == kT16Pad) return;
        const double *sl = fb + p * 5;
        const double f0 = sl[0], f1 = sl[1], f2 = sl[2], f3 = sl[3], f4 = sl[4];
        const bool neg = (code & kT16RoleB) != 0;                   // this node is the edge's b end
        a0 = neg ? a0 - f0 : a0 + f0;  a1 = neg ? a1 - f1 : a1 + f1;  a2 = neg ? a2 - f2 : a2 + f2;
        a3 = neg ? a3 - f3 : a3 + f3;  a4 = neg ? a4 - f4 : a4 + f4;
    };
#pragma unroll
    for (int k = 0; k < kGatherPre; k++)
        if (k < n_int) add_entry(gc[k]);
    for (int32_t r = kGatherPre; r < n_int; r++) add_entry(grow[r << 6]);

    if ((classes & 6) && n_bnd > 0) {
        // boundary faces need the node's own derived state again (its LDS record is gone)
        const NodeQ me = load_and_derive(q, stride, i);
        boundary_rows(me, flux_contribution(me), ff, nbr16, w, int64_t(row0) + n_int, n_bnd, lane, classes, a0, a1, a2, a3, a4);
    }

    finish_node<FUSE>(i, nel, stride, a0, a1, a2, a3, a4, fluxes, fs, min_dt, t);
}

// ------------------------------------------------------------------------------------------
// flux_half: the same three loops with every internal edge of a tile evaluated ONCE, by one of its end points.
//
// k_flux_tile streams a 26-34 byte row entry and evaluates the flux for BOTH end points of every edge; its time is the
// time to move those bytes (the indirect_rw probe through the same tiles takes 0.96 of it).  Here the plan gives every
// internal edge that touches a tile one EVALUATOR among its end points inside the tile (preprocess.hpp: half rows), so
// a tile streams one 26-byte entry per edge.  With the node records staged as in k_flux_tile, every lane walks its
// half rows — its own record in registers, the other end's from LDS, the same edge_flux as seen from this node — and
// keeps the results (at most kHalfMaxRows x 5 values) in registers; when the whole workgroup is done the records
// are dead and the results take their place in LDS (position = half row within the tile * 64 + lane, five arrays: conflict-free
// stores); every node then adds its incident edges in its own row order = the reference's accumulation order, its own
// evaluations as they are, those of its neighbours negated — which is what the reference's expressions for the other
// end evaluate to bit for bit (every term negates exactly).  An edge cut by the tile boundary is evaluated by the tile
// of each end, from identical operands.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ EdgeRow load_half_row(const uint32_t *__restrict__ code, const double *__restrict__ w3, int64_t row, int lane)
{
    EdgeRow e;
    e.code = code[(row << 6) + lane];
    const double *wr = w3 + row * (3 * kSlice) + lane;
    e.fx = wr[0]; e.fy = wr[64]; e.fz = wr[128];
    e.k = 0.0;
    return e;
}

template <bool FUSE, bool ACC>
__global__ void __launch_bounds__(kBlock, 3)
k_flux_half(// (the first 16 dwords are preloaded into SGPRs: what the prologue's first loads need)
            const double *__restrict__ q, const int32_t *__restrict__ tile_halo, uint32_t n_tiles, int32_t hr_pad_row,
            int64_t stride, int64_t nel, const int32_t *__restrict__ hr_row0, const uint32_t *__restrict__ hr_code,
            const double *__restrict__ hr_w, const uint16_t *__restrict__ hg16, const int32_t *__restrict__ slice_row0,
            const int32_t *__restrict__ rows_int, const int32_t *__restrict__ rows_bnd, int32_t pad_row,
            const uint16_t *__restrict__ nbr16, const double *__restrict__ w, FarField ff, double *__restrict__ fluxes,
            int classes, FusedStep fs)
{
    __shared__ double2 tile[kTileCap * kLdsRecD2 > kHalfLdsD2 ? kTileCap * kLdsRecD2 : kHalfLdsD2];   // (the flux terms need a little more than the records)

    PH_BEGIN();
    double min_dt = 0.0;
    if (FUSE && fs.partial_min) min_dt = block_min_of_partials(fs.partial_min, fs.n_partial);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const unsigned t = xcd_contiguous_block(blockIdx.x, n_tiles);
    const int64_t i = int64_t(t) * kTile + tid;
    const int32_t slice = __builtin_amdgcn_readfirstlane(static_cast<int32_t>(i >> 6));

    // load issue order as in k_flux_tile: halo ids, own state, the first two half rows, then the halo state by id;
    // nothing under a branch
    const int32_t *hrow = tile_halo + int64_t(t) * kHaloStride;
    const int32_t hid = hrow[tid];
    const int32_t hid2 = tid < kHaloStride - kBlock ? hrow[kBlock + tid] : -1;
    const int32_t h0 = hr_row0[slice];
    const int32_t h0_tile = hr_row0[slice & ~3];               // the tile's first half row: flux-term positions count from it
    const int32_t n_h = (classes & 1) ? hr_row0[slice + 1] - h0 : 0;
    const int32_t row0 = slice_row0[slice];
    const int32_t n_int = (classes & 1) ? rows_int[slice] : 0;
    const int32_t n_bnd = rows_bnd[slice];
    const double o0 = q[i], o1 = q[stride + i], o2 = q[2 * stride + i], o3 = q[3 * stride + i], o4 = q[4 * stride + i];
    EdgeRow e0 = load_half_row(hr_code, hr_w, n_h > 0 ? h0 : hr_pad_row, lane);
    EdgeRow e1 = load_half_row(hr_code, hr_w, n_h > 1 ? h0 + 1 : hr_pad_row, lane);
    const int64_t hnode = hid >= 0 ? int64_t(hid) : i;
    const double g0 = q[hnode], g1 = q[stride + hnode], g2 = q[2 * stride + hnode], g3 = q[3 * stride + hnode], g4 = q[4 * stride + hnode];

    // ---- phase 1: stage + derive, as k_flux_tile ----
#ifdef MGCFD_ABL_DERIVE        /* diagnostic: nothing derived (results wrong) */
    auto fake = [](double r, double mx, double my, double mz, double en) { NodeQ n; n.rho = r; n.mx = mx; n.my = my; n.mz = mz; n.en = en; n.vx = mx; n.vy = my; n.vz = mz; n.p = en; n.speed = r; n.c = r; return n; };
    const NodeQ me = fake(o0, o1, o2, o3, o4);
    lds_store_record(tile, uint32_t(tid), me);
    lds_store_record(tile, uint32_t(kTile + tid), fake(g0, g1, g2, g3, g4));
#else
    const NodeQ me = make_nodeq(o0, o1, o2, o3, o4);
    lds_store_record(tile, uint32_t(tid), me);
    lds_store_record(tile, uint32_t(kTile + tid), make_nodeq(g0, g1, g2, g3, g4));     // unconditional, see k_flux_tile
#endif
    if (hid2 >= 0) lds_store_record(tile, uint32_t(kTile + kBlock + tid), load_and_derive(q, stride, hid2));
    __syncthreads();
    PH_MARK(0);

    // ---- phase 2: this node's half rows one at a time, the two after it in flight.  (One at a time, and the node's own
    //      flux contribution recomputed per edge: the results of up to kHalfMaxRows edges wait in registers, and the
    //      kernel is bound by the bytes it moves, not by its arithmetic.) ----
    Flux5 F[kHalfMaxRows];
#pragma unroll
    for (int j = 0; j < kHalfMaxRows; j++) { F[j].d = 0.0; F[j].mx = 0.0; F[j].my = 0.0; F[j].mz = 0.0; F[j].en = 0.0; }
    uint32_t live = 0;                              // bit j: half row j of this lane holds an edge
#pragma unroll
    for (int j = 0; j < kHalfMaxRows; j++) {
        if (j < n_h) {                              // (uniform over the wave)
            EdgeRow e2 = pad_row_entry();
            if (j + 2 < kHalfMaxRows)               // compile time
                e2 = load_half_row(hr_code, hr_w, j + 2 < n_h ? h0 + j + 2 : hr_pad_row, lane);
            const uint32_t s0 = e0.code & kT16SlotMask;
            const bool v0 = s0 != kT16Pad;
            const NodeQ n0 = lds_load_record(tile, v0 ? s0 : uint32_t(tid));
            // an evaluation another node owns (the plan found no room in that node's lane): its record comes from LDS too
            NodeQ m0 = me;
            if (__builtin_expect(__any((e0.code & kHalfForeign) != 0), 0)) {
                if (e0.code & kHalfForeign) m0 = lds_load_record(tile, (e0.code >> 16) & 0xFFu);
            }
#ifdef MGCFD_ABL_EVAL          /* diagnostic: the evaluation replaced by a few adds (results wrong) */
            F[j].d = m0.rho + n0.rho + e0.fx; F[j].mx = m0.mx + n0.mx + e0.fy; F[j].my = m0.my + n0.my + e0.fz; F[j].mz = m0.mz + n0.mz + n0.c; F[j].en = m0.en + n0.en + n0.p + n0.vx + n0.speed;
#else
            F[j] = edge_flux<false>(m0, flux_contribution(m0), n0, e0);
#endif
            live |= (v0 ? 1u : 0u) << j;
            e0 = e1; e1 = e2;
        }
    }
    // the node's gather list: requested now, wanted after the hand-over
    uint32_t gc[kGatherPre];
#pragma unroll
    for (int k = 0; k < kGatherPre; k++) gc[k] = hg16[(int64_t(k < n_int ? row0 + k : pad_row) << 6) + lane];

    // ---- phase 3: the records are dead; the edge fluxes take their place ----
    double *fb = reinterpret_cast<double *>(tile);
    PH_MARK(1);
    __syncthreads();
    PH_MARK(2);
#pragma unroll
    for (int j = 0; j < kHalfMaxRows; j++) {
        if (live & (1u << j)) {
            double *slot = fb + (h0 - h0_tile + j) * kSlice + lane;
            slot[0] = F[j].d; slot[kHalfSlots] = F[j].mx; slot[2 * kHalfSlots] = F[j].my; slot[3 * kHalfSlots] = F[j].mz; slot[4 * kHalfSlots] = F[j].en;
        }
    }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0;
    if (ACC) {
        a0 = fluxes[i]; a1 = fluxes[stride + i]; a2 = fluxes[2 * stride + i];
        a3 = fluxes[3 * stride + i]; a4 = fluxes[4 * stride + i];
    }
    __syncthreads();
    PH_MARK(3);

    // ---- phase 4: every node adds its incident edges in row order ----
    auto add_entry = [&](uint32_t code) {
        const uint32_t p = code & kT16SlotMask;
        if (p == kT16Pad) return;
        const double *sl = fb + p;
        const double f0 = sl[0], f1 = sl[kHalfSlots], f2 = sl[2 * kHalfSlots], f3 = sl[3 * kHalfSlots], f4 = sl[4 * kHalfSlots];
        const bool neg = (code & kT16RoleB) != 0;                   // evaluated by the other end point
        a0 = neg ? a0 - f0 : a0 + f0;  a1 = neg ? a1 - f1 : a1 + f1;  a2 = neg ? a2 - f2 : a2 + f2;
        a3 = neg ? a3 - f3 : a3 + f3;  a4 = neg ? a4 - f4 : a4 + f4;
    };
#pragma unroll
    for (int k = 0; k < kGatherPre; k++)
        if (k < n_int) add_entry(gc[k]);
    for (int32_t r = kGatherPre; r < n_int; r++) add_entry(hg16[(int64_t(row0 + r) << 6) + lane]);

    PH_MARK(4);
    if ((classes & 6) && n_bnd > 0) {
        // (inline, as in k_flux_tile: handing the five sums to boundary_rows by reference parks two of them in scratch)
        const FluxC fm = flux_contribution(me);
        const int64_t first_bnd = int64_t(row0) + rows_int[slice];
        for (int32_t r = 0; r < n_bnd; r++) {
            const EdgeRow e = load_row<false>(nbr16, w, first_bnd + r, lane);
            const double fx = e.fx, fy = e.fy, fz = e.fz;
            if (e.code == kT16Wall && (classes & 2)) {
                // flux_boundary_kernel.elemfunc.c:37-64: pressure force only
                a0 += 0.0;
                a1 += fx * me.p;
                a2 += fy * me.p;
                a3 += fz * me.p;
                a4 += 0.0;
            } else if (e.code == kT16Far && (classes & 4)) {
                // flux_wall_kernel.elemfunc.c:51-88: average with the far-field state
                a0 += fx * (ff.var[1] + me.mx) + fy * (ff.var[2] + me.my) + fz * (ff.var[3] + me.mz);
                a4 += fx * (ff.fc_de[0] + fm.ex) + fy * (ff.fc_de[1] + fm.ey) + fz * (ff.fc_de[2] + fm.ez);
                a1 += fx * (ff.fc_mx[0] + fm.xx) + fy * (ff.fc_mx[1] + fm.xy) + fz * (ff.fc_mx[2] + fm.xz);
                a2 += fx * (ff.fc_my[0] + fm.xy) + fy * (ff.fc_my[1] + fm.yy) + fz * (ff.fc_my[2] + fm.yz);
                a3 += fx * (ff.fc_mz[0] + fm.xz) + fy * (ff.fc_mz[1] + fm.yz) + fz * (ff.fc_mz[2] + fm.zz);
            }
        }
    }

    finish_node<FUSE>(i, nel, stride, a0, a1, a2, a3, a4, fluxes, fs, min_dt, t);
    PH_MARK(5);
}

#ifdef MGCFD_ORDER_FREE
// ------------------------------------------------------------------------------------------
// flux_free (the `fast` namespace only: MGCFD_OPT_EXACT = 0, MGCFD_OPT_FLUX_VARIANT bit 6): compute_flux_edge +
// boundary + far-field faces with ORDER-FREE accumulation.
//
// Every byte-saving layout of the bit-identical kernels (edge-once tiles, indexed weights, half rows) lost what it
// saved to the hand-over that keeps the reference's summation ORDER: a second pass over LDS and two more barriers.
// north_star's bound is 1e-10, not bits, so this kernel drops the order and nothing else: a tile streams ONE 28-byte
// entry per internal edge that touches it (the half-row plan, preprocess.hpp: the evaluator of an edge inside the tile
// is one of its end points, a cut edge is evaluated by the tile of each end), the evaluating lane adds +F to its own
// node's sum in registers and — for an edge inside the tile — adds -F to the other end's sum in LDS with ds_add_f64
// (the reference's expressions for the other end evaluate to exactly -F, flux_kernel.elemfunc.c:142-189).  One barrier,
// every node adds its LDS sum to its register sum, the boundary faces follow, store.  No second pass, no ordered adds;
// the result differs from the reference's by the rounding of a differently associated sum (<= 1e-12 relative per
// launch, tests/test_gpu_order_free.py) and is not reproducible bit for bit from run to run (LDS atomics commute, their
// rounding does not).
//
// Records are 64 bytes (rho, m, E, p, |v| + c, 1/rho): with contraction allowed |v| + c may be kept as one number, and
// the velocity is three multiplications away from 1/rho — one division and two square roots per staged node instead of
// three and two, four 16-byte LDS reads per neighbour instead of six.  Quad q of slot s sits at position q ^ ((s >> 2) & 3)
// of its record, so a 16-lane ds_read_b128 group reaches all 16 quad positions of the 256-byte bank row.
// ------------------------------------------------------------------------------------------
struct NodeF { double rho, vx, vy, vz, en, p, sc; };
constexpr int kFreeRecD2 = 3;                  // double2 per record (+ one double in the array beside)

// 1/x and sqrt(x) from the hardware's estimates and Newton steps in FMAs (~1e-16 relative; the IEEE sequences hipcc emits
// under -fno-fast-math cost 15 and 20 instructions, and fp64 instructions issue at a quarter of the fp32 rate: the kernel's
// phases are bound by them, profiles/r3_free_phases.txt)
__device__ __forceinline__ double fast_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}

// sqrt(x) for x > 0 (no special cases: x = 0 gives NaN)
__device__ __forceinline__ double fast_sqrt_pos(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;                  // g -> sqrt(x), h -> 0.5 / sqrt(x) (Goldschmidt)
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    const double d = fma(-g, g, x);
    return fma(d, h, g);
}

__device__ __forceinline__ double fast_sqrt(double x)
{
    const double g = fast_sqrt_pos(x);
    // (x = 0: the estimate is infinite; +inf: the estimate is 0 and the product NaN; negative or NaN: NaN, as sqrt gives)
    return (x > 0.0 && x < __builtin_inf()) ? g : ((x == 0.0 || x == __builtin_inf()) ? x : __builtin_nan(""));
}

// What a staged node keeps (round 4): the VELOCITY instead of the momentum.  The evaluating end needs both of its neighbour
// — m for the dissipation's differences, v for the convective terms — and rho * v is three multiplications where m / rho was
// a reciprocal (five instructions) and three.
__device__ __forceinline__ NodeF make_nodef(double rho, double mx, double my, double mz, double en)
{
    NodeF n;
    const double inv = fast_rcp(rho);
    n.rho = rho; n.en = en;
    n.vx = mx * inv; n.vy = my * inv; n.vz = mz * inv;
    const double speed_sqd = n.vx * n.vx + n.vy * n.vy + n.vz * n.vz;
    n.p = (kGamma - 1.0) * (en - 0.5 * rho * speed_sqd);
    n.sc = fast_sqrt(speed_sqd) + fast_sqrt(kGamma * n.p * inv);
    return n;
}

// LDS image of a staged node: 48 bytes (rho, v, E, p) in an array of records + |v| + c in an array of its own.
// 56 bytes per node: with the tile's 10 KB of sums, 546 nodes fit the 40 KB a
// workgroup may take when FOUR share a CU.  A 12-dword record stride spreads a 16-lane ds_read_b128 group over all 16 quad
// positions of the 256-byte bank row by itself.
__device__ __forceinline__ void lds_store_nodef(double2 *rec, double *scs, uint32_t slot, const NodeF &n)
{
    double2 *r = rec + slot * kFreeRecD2;
    r[0] = make_double2(n.rho, n.vx);
    r[1] = make_double2(n.vy, n.vz);
    r[2] = make_double2(n.en, n.p);
    scs[slot] = n.sc;
}

__device__ __forceinline__ NodeF lds_load_nodef(const double2 *rec, const double *scs, uint32_t slot)
{
    const double2 *r = rec + slot * kFreeRecD2;
    const double2 a = r[0], c = r[1], d = r[2];
    NodeF n;
    n.rho = a.x; n.vx = a.y; n.vy = c.x; n.vz = c.y; n.en = d.x; n.p = d.y; n.sc = scs[slot];
    return n;
}

// what the evaluating end keeps beside its record: momentum and total enthalpy per volume
struct OwnF { double mx, my, mz, H; };
__device__ __forceinline__ OwnF make_ownf(const NodeF &n)
{
    OwnF o;
    o.mx = n.rho * n.vx; o.my = n.rho * n.vy; o.mz = n.rho * n.vz; o.H = n.en + n.p;
    return o;
}

// MINUS the flux of flux_kernel.elemfunc.c:130-161 seen from end `a` (the plan folded the b-side sign into the weights f),
// regrouped: with d_x = f . m_x = rho_x (f . v_x) the contracted flux contributions are f . Phi_x = v_x d_x + p_x f for the
// momenta and H_x (f . v_x) for the energy (cfd_loops.h:57-83), so nothing of the 3 x 3 tensors is formed.  The NEGATED
// flux is what the other end of an edge inside the tile adds to its sum (ds_add_f64 has no operand negation; the evaluating
// end subtracts, which costs nothing).  The weights of a real entry are never all zero; a padding entry's are, and it meets
// its own record as "neighbour": the length is then ~1e-150 times a difference that is exactly zero.
__device__ __forceinline__ Flux5 edge_flux_neg_f(const NodeF &a, const OwnF &oa, const NodeF &b, double fx, double fy, double fz)
{
    const double half_ewt = fast_sqrt_pos(fmax(fx * fx + fy * fy + fz * fz, 1e-300));       // :27, the plan stores f = -+0.5 e
    const double factor = (half_ewt * double(0.2f)) * (a.sc + b.sc);                          // MINUS the factor of :130-131
    const double fva = fx * a.vx + fy * a.vy + fz * a.vz;
    const double fvb = fx * b.vx + fy * b.vy + fz * b.vz;
    const double da = a.rho * fva, db = b.rho * fvb;
    const double bmx = b.rho * b.vx, bmy = b.rho * b.vy, bmz = b.rho * b.vz;
    const double ps = a.p + b.p;
    Flux5 g;
    g.d = factor * (a.rho - b.rho) - (da + db);
    g.mx = factor * (oa.mx - bmx) - a.vx * da - b.vx * db - ps * fx;
    g.my = factor * (oa.my - bmy) - a.vy * da - b.vy * db - ps * fy;
    g.mz = factor * (oa.mz - bmz) - a.vz * da - b.vz * db - ps * fz;
    g.en = factor * (a.en - b.en) - oa.H * fva - (b.en + b.p) * fvb;
    return g;
}

__device__ __forceinline__ FluxC flux_contribution_f(const NodeF &q, const OwnF &o)
{
    FluxC f;
    f.xx = q.vx * o.mx + q.p;
    f.xy = q.vx * o.my;
    f.xz = q.vx * o.mz;
    f.yy = q.vy * o.my + q.p;
    f.yz = q.vy * o.mz;
    f.zz = q.vz * o.mz + q.p;
    f.ex = q.vx * o.H;
    f.ey = q.vy * o.H;
    f.ez = q.vz * o.H;
    return f;
}

__device__ __forceinline__ void lds_add(double *p, double v) { unsafeAtomicAdd(p, v); }    // ds_add_f64

// CAP: nodes a tile may stage.  kFreeCap4 (halos of at most 290 nodes) lets four workgroups share a CU, kTileCap three.
constexpr int kFreeCap4 = 546;
static_assert(kFreeCap4 - kTile == kFreeCap4Halo, "preprocess.hpp: kFreeCap4Halo");
// LONG: some slice of the level holds more half rows than the prologue requests (the loop behind the unrolled pairs exists)
// WIDE: some tile's halo exceeds the shared table's 303 ids: the halo comes from the kernel's own table, two ids per thread
//       (tile_halo then points at it, stride kFreeHaloStride; CAP = kTile + kFreeHaloStride)
// ROLE (fused stages, round 4): -1 = the generic epilogue (finish_node: every optional path behind a run-time test, the
// step-factor partials reduced before anything else is requested); 0 ... 3 = k_flux_tile's role specialisation — 0 the first
// stage (finishes compute_step_factor: the partials are requested with the prologue and reduced under the staging barrier; the
// sweep's start state IS this stage's input, nothing of it is loaded twice), 1 a middle stage, 2 the last stage (residual or
// its squares), 3 the last stage that also leaves the next sweep's step-factor minima — with the time_step operands requested
// right behind the staging barrier, so that they arrive while the half rows are evaluated.
template <bool FUSE, bool ACC, int CAP, bool LONG, bool WIDE = false, int ROLE = -1>
__global__ void __launch_bounds__(kBlock, CAP <= kFreeCap4 ? 4 : 3)
k_flux_free(// (the first 16 dwords are preloaded into SGPRs: what the prologue's first loads need)
            const double *__restrict__ q, const int32_t *__restrict__ tile_halo, uint32_t n_tiles, int32_t hr_pad_row,
            int64_t stride, int64_t nel, const int32_t *__restrict__ hr_row0, const uint32_t *__restrict__ hr_code,
            const double *__restrict__ hr_w, const int32_t *__restrict__ slice_row0,
            const int32_t *__restrict__ rows_int, const int32_t *__restrict__ rows_bnd,
            const uint16_t *__restrict__ nbr16, const double *__restrict__ w, FarField ff, double *__restrict__ fluxes,
            int classes, FusedStep fs)
{
    __shared__ double2 tile[CAP * kFreeRecD2];
    __shared__ double scs[CAP];
    __shared__ double acc[5 * kTile];               // the sums neighbours leave for this tile's own nodes, [field][node]

    PH_BEGIN();
    constexpr bool SPEC = FUSE && ROLE >= 0;
    static_assert(ROLE < 0 || FUSE, "roles belong to the fused stages");
    double min_dt = 0.0;
    if (FUSE && !SPEC && fs.partial_min) min_dt = block_min_of_partials(fs.partial_min, fs.n_partial);
    constexpr int kPartPre = 6;                     // step-factor partials per thread held in registers (1,536 tiles)
    __shared__ double s_pm[kBlock / 64];
    double pmv[kPartPre];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const unsigned t = xcd_contiguous_block(blockIdx.x, n_tiles);
    const int64_t i = int64_t(t) * kTile + tid;
    const int32_t slice = __builtin_amdgcn_readfirstlane(static_cast<int32_t>(i >> 6));

    // load issue order as in k_flux_tile: halo ids, own state, the first two half rows, then the halo state by id;
    // nothing under a branch
    const int32_t *hrow = tile_halo + int64_t(t) * (WIDE ? kFreeHaloStride : kHaloStride);
    const int32_t hid = hrow[tid];
    const int32_t hid2 = (WIDE || tid < kHaloStride - kBlock) ? hrow[kBlock + tid] : -1;
    const int32_t h0 = hr_row0[slice];
    const int32_t n_h = (classes & 1) ? hr_row0[slice + 1] - h0 : 0;
    const int32_t n_bnd = rows_bnd[slice];
#ifdef MGCFD_ABL_FREE_OWN_AOS          /* diagnostic: the own node read as one 40-byte record, lanes 40 bytes apart (results wrong) */
    const double2 oa_ = *reinterpret_cast<const double2 *>(q + 5 * i), ob_ = *reinterpret_cast<const double2 *>(q + 5 * i + 2);
    const double o0 = oa_.x, o1 = oa_.y, o2 = ob_.x, o3 = ob_.y, o4 = q[5 * i + 4];
#else
    const double o0 = MGCFD_LD_STATE(q + i), o1 = MGCFD_LD_STATE(q + stride + i), o2 = MGCFD_LD_STATE(q + 2 * stride + i),
                 o3 = MGCFD_LD_STATE(q + 3 * stride + i), o4 = MGCFD_LD_STATE(q + 4 * stride + i);
#endif
    // EVERY half row of the lane is requested here (the plan gives a lane at most kHalfMaxRows): the row loop then waits for
    // nothing, and a workgroup has all of its tile's bytes in flight at once — what hides the memory latency is the other
    // workgroups of the CU, not a prefetch distance
    EdgeRow er[kHalfMaxRows];
    // (round 4: the halo state is requested behind the first TWO half rows and ahead of the others, so the records can be
    //  staged while the last three half rows are still on their way: 14.7 -> 14.1 us; 0, 1, 3 and 5 rows first: 14.3, 14.2,
    //  14.5, 14.7 us, profiles/r4_flux_levers.txt)
#ifdef MGCFD_EXP_FREE_ORDER
    constexpr int kRowsFirst = MGCFD_EXP_FREE_ORDER;
#else
    constexpr int kRowsFirst = 2;
#endif
#pragma unroll
    for (int j = 0; j < kRowsFirst; j++) er[j] = load_half_row(hr_code, hr_w, j < n_h ? h0 + j : hr_pad_row, lane);
    const int64_t hnode = hid >= 0 ? int64_t(hid) : i;
#if defined(MGCFD_ABL_FREE_HALO) && MGCFD_ABL_FREE_HALO == 1      /* diagnostic: the halo node read as ONE 40-byte record (results wrong) */
    const double2 ga_ = *reinterpret_cast<const double2 *>(q + 5 * hnode), gb_ = *reinterpret_cast<const double2 *>(q + 5 * hnode + 2);
    const double g0 = ga_.x, g1 = ga_.y, g2 = gb_.x, g3 = gb_.y, g4 = q[5 * hnode + 4];
#elif defined(MGCFD_ABL_FREE_HALO) && MGCFD_ABL_FREE_HALO == 2    /* diagnostic: ... as a 64-byte aligned record (results wrong) */
    const double2 ga_ = *reinterpret_cast<const double2 *>(q + 8 * (hnode >> 1)), gb_ = *reinterpret_cast<const double2 *>(q + 8 * (hnode >> 1) + 2);
    const double g0 = ga_.x, g1 = ga_.y, g2 = gb_.x, g3 = gb_.y, g4 = q[8 * (hnode >> 1) + 4];
#elif defined(MGCFD_ABL_FREE_HALO) && MGCFD_ABL_FREE_HALO == 3    /* diagnostic: no gather, the own node again (results wrong) */
    const double g0 = q[i], g1 = q[stride + i], g2 = q[2 * stride + i], g3 = q[3 * stride + i], g4 = q[4 * stride + i];
#elif defined(MGCFD_ABL_FREE_HALO) && MGCFD_ABL_FREE_HALO == 4    /* diagnostic: nothing loaded for the halo at all (results wrong) */
    const double g0 = o0 + double(hnode), g1 = o1, g2 = o2, g3 = o3, g4 = o4;
#else
    const double g0 = MGCFD_LD_STATE(q + hnode), g1 = MGCFD_LD_STATE(q + stride + hnode), g2 = MGCFD_LD_STATE(q + 2 * stride + hnode),
                 g3 = MGCFD_LD_STATE(q + 3 * stride + hnode), g4 = MGCFD_LD_STATE(q + 4 * stride + hnode);
#endif
#pragma unroll
    for (int j = kRowsFirst; j < kHalfMaxRows; j++) er[j] = load_half_row(hr_code, hr_w, j < n_h ? h0 + j : hr_pad_row, lane);
    // (WIDE: the second halo node of every thread as unconditionally as the first)
    const int64_t hnode2 = (WIDE && hid2 >= 0) ? int64_t(hid2) : i;
    double u0 = 0.0, u1 = 0.0, u2 = 0.0, u3 = 0.0, u4 = 0.0;
    if (WIDE) { u0 = q[hnode2]; u1 = q[stride + hnode2]; u2 = q[2 * stride + hnode2]; u3 = q[3 * stride + hnode2]; u4 = q[4 * stride + hnode2]; }

    if (SPEC && ROLE == 0) {                        // (requested last: wanted only at the staging barrier)
#pragma unroll
        for (int u = 0; u < kPartPre; u++) {
            const int k = threadIdx.x + u * kBlock;
            pmv[u] = fs.partial_min[k < fs.n_partial ? k : fs.n_partial - 1];     // clamped: a repeat does not change a minimum
        }
    }
#pragma unroll
    for (int f = 0; f < 5; f++) acc[f * kTile + tid] = 0.0;
    const NodeF me = make_nodef(o0, o1, o2, o3, o4);
    lds_store_nodef(tile, scs, uint32_t(tid), me);
    lds_store_nodef(tile, scs, uint32_t(kTile + tid), make_nodef(g0, g1, g2, g3, g4));          // unconditional, see k_flux_tile
    if (WIDE) lds_store_nodef(tile, scs, uint32_t(kTile + kBlock + tid), make_nodef(u0, u1, u2, u3, u4));
    else if (hid2 >= 0) {
        const int64_t h = hid2;
        lds_store_nodef(tile, scs, uint32_t(kTile + kBlock + tid), make_nodef(q[h], q[stride + h], q[2 * stride + h], q[3 * stride + h], q[4 * stride + h]));
    }
    const OwnF mo = make_ownf(me);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0;
    if (ACC) {
        a0 = fluxes[i]; a1 = fluxes[stride + i]; a2 = fluxes[2 * stride + i];
        a3 = fluxes[3 * stride + i]; a4 = fluxes[4 * stride + i];
    }
    if (SPEC && ROLE == 0) {
        double pm = pmv[0];
#pragma unroll
        for (int u = 1; u < kPartPre; u++) pm = fmin(pm, pmv[u]);
        for (int k = threadIdx.x + kPartPre * kBlock; k < fs.n_partial; k += kBlock) pm = fmin(pm, fs.partial_min[k]);
        pm = wave_min(pm);
        if (lane == 0) s_pm[tid >> 6] = pm;
    }
    PH_MARK(0);
    __syncthreads();
    PH_MARK(1);
    // role-specialised stages: the time_step operands go out behind the first pair of half rows (whose registers they take)
    // and arrive under the others
    double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0, r4 = 0.0, sfv = 0.0;
    auto request_operands = [&](int part) {        // part 0 behind the first pair, part 1 behind the second (the register budget of four waves per SIMD)
        if (ROLE == 0) {
            // the first stage's input IS the sweep's start state (the launcher checks q == old_variables): rho and E are in the
            // node's record, the momenta are rho v again (this kernel's results are not the reference's bits anyway)
            if (part == 0) { r0 = me.rho; r1 = mo.mx; r2 = mo.my; r3 = mo.mz; r4 = me.en; sfv = fs.volumes[i]; }
        } else if (part == 0) {
            r0 = fs.old_variables[i]; r1 = fs.old_variables[stride + i]; r2 = fs.old_variables[2 * stride + i];
            r3 = fs.old_variables[3 * stride + i];
        } else {
            r4 = fs.old_variables[4 * stride + i];
            sfv = fs.step_factors[i];
        }
    };
    if (SPEC && n_h <= 0) request_operands(0);      // (uniform over the wave: a slice without half rows)
    if (SPEC && n_h <= 2) request_operands(1);

    // ---- this lane's half rows, two at a time (independent arithmetic; a half row the slice does not have was read from the
    //      padding row: zero weights, nothing added) ----
    auto eval = [&](const EdgeRow &e0, bool may_be_foreign) {
        const uint32_t s = e0.code & kT16SlotMask;
        const bool v = s != kT16Pad;
        const NodeF ot = lds_load_nodef(tile, scs, v ? s : uint32_t(tid));
        Flux5 G;                                    // MINUS the edge's flux as its evaluating end sees it
        bool mine = true;
        if (may_be_foreign) {
            // an evaluation another node owns (the plan found no room in that node's lane): its record comes from LDS too,
            // and its share goes to its LDS sum
            const bool foreign = (e0.code & kHalfForeign) != 0;
            const uint32_t own = (e0.code >> 16) & 0xFFu;
            const NodeF m0 = foreign ? lds_load_nodef(tile, scs, own) : me;
            G = edge_flux_neg_f(m0, make_ownf(m0), ot, e0.fx, e0.fy, e0.fz);
            if (foreign && v) {
                lds_add(&acc[own], -G.d); lds_add(&acc[kTile + own], -G.mx); lds_add(&acc[2 * kTile + own], -G.my);
                lds_add(&acc[3 * kTile + own], -G.mz); lds_add(&acc[4 * kTile + own], -G.en);
            }
            mine = !foreign;
        } else {
            G = edge_flux_neg_f(me, mo, ot, e0.fx, e0.fy, e0.fz);
        }
        // (padding: G = 0)
        if (mine) { a0 -= G.d; a1 -= G.mx; a2 -= G.my; a3 -= G.mz; a4 -= G.en; }
        if (v && (e0.code & kHalfMirror)) {
            // the other end lies in this tile and does not evaluate the edge itself: it gets -F
            lds_add(&acc[s], G.d); lds_add(&acc[kTile + s], G.mx); lds_add(&acc[2 * kTile + s], G.my);
            lds_add(&acc[3 * kTile + s], G.mz); lds_add(&acc[4 * kTile + s], G.en);
        }
    };
#pragma unroll
    for (int j = 0; j < kHalfMaxRows; j += 2) {
        if (j >= n_h) break;                        // (uniform over the wave)
        const bool two = j + 1 < kHalfMaxRows;      // (compile time)
        const bool any_foreign = ((er[j].code | (two ? er[j + 1 < kHalfMaxRows ? j + 1 : j].code : 0u)) & kHalfForeign) != 0;
        if (__builtin_expect(__any(any_foreign), 0)) {
            eval(er[j], true);
            if (two) eval(er[j + 1 < kHalfMaxRows ? j + 1 : j], true);
        } else {
            eval(er[j], false);
            if (two) eval(er[j + 1 < kHalfMaxRows ? j + 1 : j], false);
        }
        if (SPEC && j == 0) request_operands(0);
        if (SPEC && j == 2) request_operands(1);
    }
    // a slice with more half rows than the prologue requests (tetrahedral regions, hubs: the plan spreads a high-degree node's
    // evaluations over the tile's lanes, preprocess.hpp kFreeMaxRows): the rest one at a time, two in flight
    if (LONG && n_h > kHalfMaxRows) {               // (uniform over the wave)
        EdgeRow x0 = load_half_row(hr_code, hr_w, h0 + kHalfMaxRows, lane);
        EdgeRow x1 = load_half_row(hr_code, hr_w, kHalfMaxRows + 1 < n_h ? h0 + kHalfMaxRows + 1 : hr_pad_row, lane);
        for (int32_t j = kHalfMaxRows; j < n_h; j++) {
            const EdgeRow x2 = load_half_row(hr_code, hr_w, j + 2 < n_h ? h0 + j + 2 : hr_pad_row, lane);
            if (__any((x0.code & kHalfForeign) != 0)) eval(x0, true);
            else eval(x0, false);
            x0 = x1; x1 = x2;
        }
    }
    PH_MARK(2);
    __syncthreads();
    PH_MARK(3);
    a0 += acc[tid]; a1 += acc[kTile + tid]; a2 += acc[2 * kTile + tid]; a3 += acc[3 * kTile + tid]; a4 += acc[4 * kTile + tid];

    if ((classes & 6) && n_bnd > 0) {
        const int32_t row0 = slice_row0[slice];
        const int64_t first_bnd = int64_t(row0) + rows_int[slice];
        const FluxC fm = flux_contribution_f(me, mo);
        for (int32_t r = 0; r < n_bnd; r++) {
            const EdgeRow e = load_row<false>(nbr16, w, first_bnd + r, lane);
            const double fx = e.fx, fy = e.fy, fz = e.fz;
            if (e.code == kT16Wall && (classes & 2)) {
                // flux_boundary_kernel.elemfunc.c:37-64: pressure force only
                a1 += fx * me.p;
                a2 += fy * me.p;
                a3 += fz * me.p;
            } else if (e.code == kT16Far && (classes & 4)) {
                // flux_wall_kernel.elemfunc.c:51-88: average with the far-field state
                a0 += fx * (ff.var[1] + mo.mx) + fy * (ff.var[2] + mo.my) + fz * (ff.var[3] + mo.mz);
                a4 += fx * (ff.fc_de[0] + fm.ex) + fy * (ff.fc_de[1] + fm.ey) + fz * (ff.fc_de[2] + fm.ez);
                a1 += fx * (ff.fc_mx[0] + fm.xx) + fy * (ff.fc_mx[1] + fm.xy) + fz * (ff.fc_mx[2] + fm.xz);
                a2 += fx * (ff.fc_my[0] + fm.xy) + fy * (ff.fc_my[1] + fm.yy) + fz * (ff.fc_my[2] + fm.yz);
                a3 += fx * (ff.fc_mz[0] + fm.xz) + fy * (ff.fc_mz[1] + fm.yz) + fz * (ff.fc_mz[2] + fm.zz);
            }
        }
    }
    if (!SPEC) {
        finish_node<FUSE>(i, nel, stride, a0, a1, a2, a3, a4, fluxes, fs, min_dt, t);
        PH_MARK(4);
        return;
    }
    // ---- fused time_step, role-specialised (k_flux_tile's epilogue: the same operations as k_time_step, cfd_loops.cpp:241-268) ----
    double sf_next = __longlong_as_double(0x7FF0000000000000LL);          // +inf: lanes past nel
    double ss = 0.0;                                                      // this node's share of the residual sum of squares
    if (i < nel) {
        double sf = sfv;
        if (ROLE == 0) {                            // finish compute_step_factor (cfd_loops.cpp:137-156)
            min_dt = s_pm[0];
            for (int wv = 1; wv < kBlock / 64; wv++) min_dt = fmin(min_dt, s_pm[wv]);
            sf = min_dt / sfv;                      // sfv holds the volume
            MGCFD_ST_STAGE(fs.step_factors + i, sf);
        }
        const double factor = sf / fs.rk_div;
        const double rho = r0 + factor * a0, mx = r1 + factor * a1, my = r2 + factor * a2, mz = r3 + factor * a3,
                     en = r4 + factor * a4;
        store_conserved(fs.q_out, stride, i, rho, mx, my, mz, en);
        if (ROLE >= 2) {                            // last stage: residual (validation.cpp:77-89)
            const double d0 = rho - r0, d1 = mx - r1, d2 = my - r2, d3 = mz - r3, d4 = en - r4;
            if (fs.residuals) {                      // (null: the caller writes it on demand, solver.cpp settle_residuals)
                MGCFD_ST_STAGE(fs.residuals + i, d0); MGCFD_ST_STAGE(fs.residuals + stride + i, d1); MGCFD_ST_STAGE(fs.residuals + 2 * stride + i, d2);
                MGCFD_ST_STAGE(fs.residuals + 3 * stride + i, d3); MGCFD_ST_STAGE(fs.residuals + 4 * stride + i, d4);
            }
            if (fs.sumsq_partial) ss = (((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3) + d4 * d4;
        }
        if (fs.check) {
            const bool finite = isfinite(rho) && isfinite(mx) && isfinite(my) && isfinite(mz) && isfinite(en);
            int code = 0;
            if (!finite) code = 1;
            else if (rho < 0.0) code = 2;
            else if (en < 0.0) code = 3;
            if (code) atomicMin(fs.err, err_key(fs.check, fs.old_of_new[i], code));
        }
        if (ROLE == 3) {                            // look-ahead: the next sweep's compute_step_factor starts from the state just produced
            const Derived d = derive(rho, mx, my, mz, en);
            const double dt = fs.cbrt_vol[i] / (d.speed + d.c);          // k_step_factor_local
            sf_next = 0.5 * dt;
        }
    }
    if (ROLE == 3 || (ROLE == 2 && fs.sumsq_partial)) {                // uniform: every thread of the workgroup takes part
        __shared__ double s_next[2][kBlock / 64];
        if (ROLE == 3) sf_next = wave_min(sf_next);
        if (fs.sumsq_partial) ss = wave_sum(ss);
        if ((threadIdx.x & 63) == 0) { s_next[0][threadIdx.x >> 6] = sf_next; s_next[1][threadIdx.x >> 6] = ss; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double m = s_next[0][0], sum = s_next[1][0];
            for (int wv = 1; wv < kBlock / 64; wv++) { m = fmin(m, s_next[0][wv]); sum += s_next[1][wv]; }
            if (ROLE == 3) fs.next_partial_min[t] = m;
            if (fs.sumsq_partial) fs.sumsq_partial[t] = sum;
        }
    }
    PH_MARK(4);
}
#endif // MGCFD_ORDER_FREE

// ------------------------------------------------------------------------------------------
// Two-phase ("fission") design point, MGCFD_OPT_FLUX_VARIANT bit 2 — the GPU form of the
// reference's FLUX_FISSION build (flux_kernel.elemfunc.c:193-204 + update_edges,
// cfd_loops.cpp:159-213): phase 1 evaluates every internal edge once, edge-parallel, and writes its
// five fluxes to memory; phase 2 is a node-centred sum of each node's incident edges in row order
// (+F at the a end, -F at the b end) followed by the boundary rows.  Bit-identical to the other
// variants; kept to put a number on the scatter-strategy choice.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_fission_edge_flux(int64_t n_edges, int64_t n_edges_pad, int64_t stride, const double *__restrict__ q,
                    const int32_t *__restrict__ fe_ab, const double *__restrict__ fe_w, double *__restrict__ eflux)
{
    const int64_t e = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    if (e >= n_edges) return;
    const int64_t a = fe_ab[e], b = fe_ab[n_edges + e];
    EdgeRow er;
    er.code = 0;                                                     // a-side form
    er.fx = fe_w[e]; er.fy = fe_w[n_edges + e]; er.fz = fe_w[2 * n_edges + e]; er.k = fe_w[3 * n_edges + e];
    const NodeQ A = load_and_derive(q, stride, a), B = load_and_derive(q, stride, b);
    const Flux5 f = edge_flux<true>(A, flux_contribution(A), B, er);
    eflux[e] = f.d; eflux[n_edges_pad + e] = f.mx; eflux[2 * n_edges_pad + e] = f.my;
    eflux[3 * n_edges_pad + e] = f.mz; eflux[4 * n_edges_pad + e] = f.en;
}

template <bool ACC>
__global__ void __launch_bounds__(kBlock)
k_fission_node_sum(int64_t nel, int64_t stride, int64_t n_edges_pad, const double *__restrict__ q,
                   const int32_t *__restrict__ slice_row0, const int32_t *__restrict__ rows_int,
                   const int32_t *__restrict__ rows_bnd, const int32_t *__restrict__ row_edge,
                   const uint16_t *__restrict__ nbr16, const double *__restrict__ w,
                   const double *__restrict__ eflux, FarField ff, double *__restrict__ fluxes, int classes)
{
    const unsigned blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int64_t i = blk * int64_t(kBlock) + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int32_t slice = __builtin_amdgcn_readfirstlane(static_cast<int32_t>(i >> 6));
    const int32_t row0 = slice_row0[slice];
    const int32_t n_int = rows_int[slice];
    const int32_t n_bnd = rows_bnd[slice];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0;
    if (ACC) {
        a0 = fluxes[i]; a1 = fluxes[stride + i]; a2 = fluxes[2 * stride + i];
        a3 = fluxes[3 * stride + i]; a4 = fluxes[4 * stride + i];
    }
    if (classes & 1) {
        const int32_t *re = row_edge + (int64_t(row0) << 6) + lane;
        for (int32_t r = 0; r < n_int; r++) {
            const int32_t code = re[int64_t(r) << 6];
            if (code == -1) continue;                                // ELL padding
            const int64_t e = code & 0x7FFFFFFF;
            const double f0 = eflux[e], f1 = eflux[n_edges_pad + e], f2 = eflux[2 * n_edges_pad + e],
                         f3 = eflux[3 * n_edges_pad + e], f4 = eflux[4 * n_edges_pad + e];
            if (code < 0) { a0 -= f0; a1 -= f1; a2 -= f2; a3 -= f3; a4 -= f4; }      // this node is the b end
            else { a0 += f0; a1 += f1; a2 += f2; a3 += f3; a4 += f4; }
        }
    }
    if ((classes & 6) && n_bnd > 0) {
        const NodeQ me = load_and_derive(q, stride, i);
        boundary_rows(me, flux_contribution(me), ff, nbr16, w, int64_t(row0) + n_int, n_bnd, lane, classes, a0, a1, a2, a3, a4);
    }
    if (i < nel) {
        fluxes[i] = a0; fluxes[stride + i] = a1; fluxes[2 * stride + i] = a2;
        fluxes[3 * stride + i] = a3; fluxes[4 * stride + i] = a4;
    }
}

// ------------------------------------------------------------------------------------------
// indirect_rw (indirect_rw_kernel.elemfunc.c:4-94) in gather form: the reference's "same data
// movement, minimal arithmetic" probe.  a-side gets q_b + (ex, ez, 0, 0, ey); b-side gets q_a.
// The plan stores -0.5*e (a side), so e = -2*w exactly.  Neighbour state straight from HBM/L2.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_indirect_rw(int64_t nel, int64_t stride, const double *__restrict__ q, const int32_t *__restrict__ slice_row0,
              const int32_t *__restrict__ rows_int, const int32_t *__restrict__ nbr,
              const double *__restrict__ w, double *__restrict__ fluxes)
{
    const unsigned blk = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int64_t i = blk * int64_t(kBlock) + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int32_t slice = __builtin_amdgcn_readfirstlane(static_cast<int32_t>(i >> 6));
    const int32_t row0 = slice_row0[slice];
    const int32_t n_int = rows_int[slice];
    double a0 = fluxes[i], a1 = fluxes[stride + i], a2 = fluxes[2 * stride + i],
           a3 = fluxes[3 * stride + i], a4 = fluxes[4 * stride + i];
    for (int32_t r = 0; r < n_int; r++) {
        const int64_t row = int64_t(row0) + r;
        const int32_t code = nbr[(row << 6) + lane];
        if (code < 0) continue;
        const int64_t j = code & kIdMask;
        const double o0 = q[j], o1 = q[stride + j], o2 = q[2 * stride + j], o3 = q[3 * stride + j], o4 = q[4 * stride + j];
        if (code & kRoleB) {
            a0 += o0; a1 += o1; a2 += o2; a3 += o3; a4 += o4;
        } else {
            const double *wr = w + (row << 8) + lane;
            a0 += o0 + (-2.0 * wr[0]);
            a1 += o1 + (-2.0 * wr[128]);
            a2 += o2;
            a3 += o3;
            a4 += o4 + (-2.0 * wr[64]);
        }
    }
    if (i < nel) {
        fluxes[i] = a0; fluxes[stride + i] = a1; fluxes[2 * stride + i] = a2;
        fluxes[3 * stride + i] = a3; fluxes[4 * stride + i] = a4;
    }
}

// ------------------------------------------------------------------------------------------
// indirect_rw through the LDS tiles: the probe the reference keeps to bound compute_flux_edge from above
// (indirect_rw_loop.cpp:8-10, "same data movement, minimal arithmetic").  Here "same data movement" means the flux
// kernel's own: the same prologue (halo ids, own and halo state into 96-byte LDS records — the derived fields filled
// with copies, nothing is derived), the same incidence rows two at a time with the same prefetch, every byte the
// flux kernel loads is loaded (weights of both roles and the length factor included; an empty asm consumes what the
// probe's arithmetic does not use), the same stores.  Arithmetic: indirect_rw_kernel.elemfunc.c:41-55 in gather form.
// Levels whose tiles leave halo nodes outside LDS or have long rows use k_indirect_rw above.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void keep_alive(double x) { asm volatile("" ::"v"(x)); }

template <bool LOADK>
__global__ void __launch_bounds__(kBlock, 3)
k_indirect_rw_tile(const double *__restrict__ q, const int32_t *__restrict__ tile_halo, uint32_t n_tiles, int32_t pad_row,
                   int64_t stride, int64_t nel, const int32_t *__restrict__ slice_row0, const int32_t *__restrict__ rows_int,
                   const uint16_t *__restrict__ nbr16, const double *__restrict__ w, double *__restrict__ fluxes)
{
    __shared__ double2 tile[kTileCap * kLdsRecD2];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const unsigned t = xcd_contiguous_block(blockIdx.x, n_tiles);
    const int64_t i = int64_t(t) * kTile + tid;
    const int32_t slice = __builtin_amdgcn_readfirstlane(static_cast<int32_t>(i >> 6));
    const int32_t *hrow = tile_halo + int64_t(t) * kHaloStride;
    const int32_t hid = hrow[tid];
    const int32_t hid2 = tid < kHaloStride - kBlock ? hrow[kBlock + tid] : -1;
    const int32_t row0 = slice_row0[slice];
    const int32_t n_int = rows_int[slice];
#if defined(MGCFD_ABL_NO_HALO)      /* diagnostic: the halo state not gathered: the dependent ids -> state chain is gone (results wrong) */
    const int64_t hnode = hid >= 0 ? i : i;
#elif defined(MGCFD_ABL_HALO_COALESCED)   /* diagnostic: the chain ids -> state kept, but the gather made contiguous (results wrong) */
    const int64_t hnode = hid >= 0 ? ((int64_t(__builtin_amdgcn_readfirstlane(hid)) & ~int64_t(255)) + tid) : i;
#else
    const int64_t hnode = hid >= 0 ? int64_t(hid) : i;
#endif
    const double o0 = q[i], o1 = q[stride + i], o2 = q[2 * stride + i], o3 = q[3 * stride + i], o4 = q[4 * stride + i];
    const double g0 = q[hnode], g1 = q[stride + hnode], g2 = q[2 * stride + hnode], g3 = q[3 * stride + hnode], g4 = q[4 * stride + hnode];
    EdgeRow e0 = load_row<LOADK>(nbr16, w, n_int > 0 ? row0 : pad_row, lane);
    EdgeRow e1 = load_row<LOADK>(nbr16, w, n_int > 1 ? row0 + 1 : pad_row, lane);
    auto record = [](double r, double mx, double my, double mz, double en) {
        NodeQ n;
        n.rho = r; n.mx = mx; n.my = my; n.mz = mz; n.en = en;
        n.vx = r; n.vy = mx; n.vz = my; n.p = mz; n.speed = en; n.c = r;      // (copies: the record keeps its 96 bytes)
        return n;
    };
    lds_store_record(tile, uint32_t(tid), record(o0, o1, o2, o3, o4));
    lds_store_record(tile, uint32_t(kTile + tid), record(g0, g1, g2, g3, g4));
    if (hid2 >= 0) {
        const int64_t h = hid2;
        lds_store_record(tile, uint32_t(kTile + kBlock + tid), record(q[h], q[stride + h], q[2 * stride + h], q[3 * stride + h], q[4 * stride + h]));
    }
    // fluxes += ... (indirect_rw_kernel.elemfunc.c:84-94)
    double a0 = fluxes[i], a1 = fluxes[stride + i], a2 = fluxes[2 * stride + i], a3 = fluxes[3 * stride + i], a4 = fluxes[4 * stride + i];
    __syncthreads();
    auto entry = [&](const EdgeRow &e) {
        const uint32_t s = e.code & kT16SlotMask;
        const bool v = s != kT16Pad;
#if defined(MGCFD_ABL_LDS_OWN)      /* diagnostic: every lane reads its own record: no bank conflicts (results wrong) */
        const NodeQ n = lds_load_record(tile, uint32_t(tid));
#else
        const NodeQ n = lds_load_record(tile, v ? s : uint32_t(tid));
#endif
        keep_alive(n.vx); keep_alive(n.vy); keep_alive(n.vz); keep_alive(n.p); keep_alive(n.speed); keep_alive(n.c);
        keep_alive(e.fx); keep_alive(e.fy); keep_alive(e.fz);
        if (LOADK) keep_alive(e.k);
        if (!v) return;
        if (e.code & kT16RoleB) {                                  // this node is the edge's b end: += q_a (:50-54)
            a0 += n.rho; a1 += n.mx; a2 += n.my; a3 += n.mz; a4 += n.en;
        } else {                                                    // a end: += q_b + (ex, ez, 0, 0, ey) (:41-45); the plan stores -0.5*e
            a0 += n.rho + (-2.0 * e.fx);
            a1 += n.mx + (-2.0 * e.fz);
            a2 += n.my;
            a3 += n.mz;
            a4 += n.en + (-2.0 * e.fy);
        }
    };
    int32_t r = 0;
    for (; r + 2 < n_int; r += 2) {
        const EdgeRow e2 = load_row<LOADK>(nbr16, w, row0 + r + 2, lane);
        const EdgeRow e3 = load_row<LOADK>(nbr16, w, r + 3 < n_int ? row0 + r + 3 : pad_row, lane);
        entry(e0); entry(e1);
        e0 = e2; e1 = e3;
    }
    if (r < n_int) { entry(e0); entry(e1); }
    if (i < nel) {
        fluxes[i] = a0; fluxes[stride + i] = a1; fluxes[2 * stride + i] = a2;
        fluxes[3 * stride + i] = a3; fluxes[4 * stride + i] = a4;
    }
}

// ------------------------------------------------------------------------------------------
// time_step (cfd_loops.cpp:241-268): variables = old + sf/(RK+1-j) * fluxes ; fluxes = 0.
// Fused options (same operations, fewer passes over memory):
//   * partial_min != nullptr: this is the first stage after compute_step_factor's first half —
//     finish it here: min over the workgroups' partial minima, then
//     step_factors[i] = min_dt / volumes[i] (cfd_loops.cpp:137-156);
//   * zero_fluxes == 0: leave fluxes[] stale; the caller treats the array as logically zero and
//     the next flux launch overwrites it (saves the 40 B/node of zero stores);
//   * residuals != nullptr: last stage — residuals = variables - old_variables (validation.cpp:77-89);
//   * check: raise the check_for_invalid_variables flag (validation.cpp:107-138),
//     err = (smallest offending ORIGINAL cell id << 8) | code.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_time_step(int64_t nel, int64_t stride, double rk_div, double *__restrict__ step_factors,
            double *__restrict__ fluxes, const double *__restrict__ old_variables, double *__restrict__ q,
            const int32_t *__restrict__ old_of_new, unsigned long long *__restrict__ err, int check,
            const double *__restrict__ partial_min, int n_partial, const double *__restrict__ volumes,
            double *__restrict__ residuals, int zero_fluxes)
{
    const int64_t i = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    double min_dt = 0.0;
    if (partial_min) min_dt = block_min_of_partials(partial_min, n_partial);   // all threads take part
    if (i >= nel) return;
    double sf;
    if (partial_min) {
        sf = min_dt / volumes[i];
        step_factors[i] = sf;
    } else {
        sf = step_factors[i];
    }
    const double factor = sf / rk_div;
    const double r0 = old_variables[i], r1 = old_variables[stride + i], r2 = old_variables[2 * stride + i],
                 r3 = old_variables[3 * stride + i], r4 = old_variables[4 * stride + i];
    const double rho = r0 + factor * fluxes[i];
    const double mx = r1 + factor * fluxes[stride + i];
    const double my = r2 + factor * fluxes[2 * stride + i];
    const double mz = r3 + factor * fluxes[3 * stride + i];
    const double en = r4 + factor * fluxes[4 * stride + i];
    store_conserved(q, stride, i, rho, mx, my, mz, en);
    if (zero_fluxes) {
        fluxes[i] = 0.0; fluxes[stride + i] = 0.0; fluxes[2 * stride + i] = 0.0;
        fluxes[3 * stride + i] = 0.0; fluxes[4 * stride + i] = 0.0;
    }
    if (residuals) {
        residuals[i] = rho - r0; residuals[stride + i] = mx - r1; residuals[2 * stride + i] = my - r2;
        residuals[3 * stride + i] = mz - r3; residuals[4 * stride + i] = en - r4;
    }
    if (check) {
        const bool finite = isfinite(rho) && isfinite(mx) && isfinite(my) && isfinite(mz) && isfinite(en);
        int code = 0;
        if (!finite) code = 1;
        else if (rho < 0.0) code = 2;
        else if (en < 0.0) code = 3;
        if (code) {
            atomicMin(err, err_key(check, old_of_new[i], code));
        }
    }
}

// check_for_invalid_variables as a standalone sweep
__global__ void __launch_bounds__(kBlock)
k_check_invalid(int64_t nel, int64_t stride, const double *__restrict__ q, const int32_t *__restrict__ old_of_new,
                unsigned long long *__restrict__ err)
{
    const int64_t i = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    if (i >= nel) return;
    const double v0 = q[i], v1 = q[stride + i], v2 = q[2 * stride + i], v3 = q[3 * stride + i], v4 = q[4 * stride + i];
    const bool finite = isfinite(v0) && isfinite(v1) && isfinite(v2) && isfinite(v3) && isfinite(v4);
    int code = 0;
    if (!finite) code = 1;
    else if (v0 < 0.0) code = 2;
    else if (v4 < 0.0) code = 3;
    if (code) atomicMin(err, err_key(1, old_of_new[i], code));
}

// residual (validation.cpp:77-89), flat over the 5 padded fields
__global__ void __launch_bounds__(kBlock)
k_residual(int64_t n, const double *__restrict__ old_variables, const double *__restrict__ variables,
           double *__restrict__ residuals)
{
    const int64_t k = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    if (k < n) residuals[k] = variables[k] - old_variables[k];
}

// sum of squares for calc_rms (validation.cpp:91-105) over the nel real entries of 5 fields.
// Tree order differs from the reference's serial sum; the value is only ever printed with %.3e.
__global__ void __launch_bounds__(kBlock)
k_sumsq(int64_t nel, int64_t stride, const double *__restrict__ x, double *__restrict__ partial,
        const int32_t *__restrict__ old_of_new, int64_t n_owned)
{
    __shared__ double s[kBlock / 64];
    double acc = 0.0;
    for (int64_t i = blockIdx.x * int64_t(kBlock) + threadIdx.x; i < nel; i += int64_t(gridDim.x) * kBlock) {
        if (old_of_new && old_of_new[i] >= n_owned) continue;        // ghost of a partitioned level: counted by its owner
        for (int f = 0; f < 5; f++) { const double v = x[f * stride + i]; acc += v * v; }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kBlock / 64; w++) t += s[w];
        partial[blockIdx.x] = t;
    }
}

// One workgroup adds up n partial sums (always in this order) and may append the total to the rms history.
__device__ __forceinline__ void block_sum_partials(const SumTask &task)
{
    __shared__ double s[kBlock / 64];
    double acc = 0.0;
    for (int k = threadIdx.x; k < task.n; k += kBlock) acc += task.partial[k];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kBlock / 64; w++) t += s[w];
        task.out[0] = t;
        if (task.ring) {
            const int c = *task.count;
            if (c < task.cap) task.ring[c] = t;
            *task.count = c + 1;
        }
    }
}

__global__ void __launch_bounds__(kBlock)
k_sum_partials(SumTask task)
{
    block_sum_partials(task);
}

// ------------------------------------------------------------------------------------------
// Halo exchange of a partitioned level: pack n nodes' 5 values into a contiguous [n][5] message
// (what an RCCL send/recv moves) and unpack a received message into the ghost nodes.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_halo_pack(int64_t n, int64_t stride, const int32_t *__restrict__ idx, const double *__restrict__ field, double *__restrict__ msg)
{
    const int64_t k = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    if (k >= n) return;
    const int64_t i = idx[k];
    for (int f = 0; f < 5; f++) msg[k * 5 + f] = field[f * stride + i];
}

__global__ void __launch_bounds__(kBlock)
k_halo_unpack(int64_t n, int64_t stride, const int32_t *__restrict__ idx, const double *__restrict__ msg, double *__restrict__ field)
{
    const int64_t k = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    if (k >= n) return;
    const int64_t i = idx[k];
    for (int f = 0; f < 5; f++) field[f * stride + i] = msg[k * 5 + f];
}

// The message of a partitioned level stored straight into the neighbours' ghost slots: slot k of the message is node idx[k]
// here and node target[k] in the numbering of the peer whose segment k lies in.  One launch replaces pack, the copies (or
// send/receive) and the peers' unpack launches; the stores cross xGMI (peer access) or stay on the device.
__global__ void __launch_bounds__(kBlock)
k_halo_push(int64_t n, int64_t stride, const int32_t *__restrict__ idx, const int32_t *__restrict__ target,
            const double *__restrict__ field, PushPeers peers)
{
    const int64_t k = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    if (k >= n) return;
    double *dst = peers.base[0];
    int64_t ps = peers.stride[0];
#pragma unroll
    for (int p = 1; p < kMaxPushPeers; p++)                          // (selects, no indexed access to the argument block)
        if (p < peers.n && k >= peers.first[p]) { dst = peers.base[p]; ps = peers.stride[p]; }
    const int64_t i = idx[k], g = target[k];
    for (int f = 0; f < 5; f++) dst[f * ps + g] = field[f * stride + i];
}

// The same between PROCESSES (the peers' buffers opened through HIP IPC): the push also tells every peer that the message is
// complete, by a sequence number in a word of the peer's memory.  Every wave drains its stores, the workgroup counts itself
// off (device-scope ticket), and the workgroup that counts last raises the flags behind a system-scope release — so a peer
// that reads the number finds the whole message.  The receiver does not poll inside a compute kernel: k_flags_wait is a
// launch of its own in front of the boundary tiles (one wave; it gives up after about two seconds and says so), and the
// boundary launch behind it starts with the acquire every launch starts with.
__global__ void __launch_bounds__(kBlock)
k_halo_push_flags(int64_t n, int64_t stride, const int32_t *__restrict__ idx, const int32_t *__restrict__ target,
                  const double *__restrict__ field, PushPeers peers, PushFlags flags, unsigned *__restrict__ ticket)
{
    const int64_t k = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    if (k < n) {
        double *dst = peers.base[0];
        int64_t ps = peers.stride[0];
#pragma unroll
        for (int p = 1; p < kMaxPushPeers; p++)
            if (p < peers.n && k >= peers.first[p]) { dst = peers.base[p]; ps = peers.stride[p]; }
        const int64_t i = idx[k], g = target[k];
        for (int f = 0; f < 5; f++) dst[f * ps + g] = field[f * stride + i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        const unsigned done = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // (the next push starts from zero again)
            __threadfence_system();
#pragma unroll
            for (int p = 0; p < kMaxPushPeers; p++)
                if (p < flags.n) __hip_atomic_store(flags.flag[p], flags.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// wait until the flag words [row][slot] of the given source ranks have reached `value` (see above); *timed_out counts the waits that gave up
__global__ void k_flags_wait(const unsigned long long *__restrict__ flags, FlagRows rows, int slot, unsigned long long value,
                             int *__restrict__ timed_out)
{
    const int p = threadIdx.x;
    int row = rows.row[0];
#pragma unroll
    for (int q = 1; q < kMaxIpcRanks; q++) if (q == p) row = rows.row[q];         // (selects, no indexed access to the argument block)
    if (p >= rows.n) return;
    const unsigned long long *w = flags + row * 4 + slot;
    const unsigned long long t0 = wall_clock64();                       // (100 MHz)
    while (__hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < value) {
        __builtin_amdgcn_s_sleep(64);
        if (wall_clock64() - t0 > 200000000ull) { atomicAdd(timed_out, 1); break; }
    }
}

// this rank's time-step minimum into every rank's array, then their flags (one lane per destination rank)
__global__ void k_min_publish(const double *__restrict__ my_min, MinPublish mp)
{
    const int p = threadIdx.x;
    double *dst = mp.mins[0];
    unsigned long long *flag = mp.flag[0];
#pragma unroll
    for (int q = 1; q < kMaxIpcRanks; q++) if (q == p) { dst = mp.mins[q]; flag = mp.flag[q]; }
    if (p >= mp.world) return;
    const double v = *my_min;
    __hip_atomic_store(dst + mp.parity * kMaxIpcRanks + mp.me, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (p == mp.me) return;
    __threadfence_system();
    __hip_atomic_store(flag, mp.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One multigrid level per rank: the restricted variables arrive from the rank that holds the finer level as a whole
// [5][stride] array.  mg_restrict leaves a coarse node WITHOUT children at its old value (mg_loops.cpp:63-78,174-189),
// and only the rank that sweeps the coarse level has that value: take the message for nodes with children only.
__global__ void __launch_bounds__(kBlock)
k_accept_restricted(int64_t nel_coarse, int64_t stride_coarse, const int32_t *__restrict__ child_ptr,
                    const double *__restrict__ src, double *__restrict__ coarse_q)
{
    const int64_t c = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    if (c >= nel_coarse || child_ptr[c + 1] == child_ptr[c]) return;
    for (int f = 0; f < 5; f++) coarse_q[f * stride_coarse + c] = src[f * stride_coarse + c];
}

// In-process groups (several solvers of one process, peer access between their devices): the all-reduce(MIN) of the
// time step is every rank reading the others' scalars directly (8-byte loads over xGMI) behind their events.
__global__ void k_min_over_peers(const double *const *__restrict__ scalars, int n, double *__restrict__ out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double m = scalars[0][0];
        for (int r = 1; r < n; r++) m = fmin(m, scalars[r][0]);
        out[0] = m;
    }
}

// rms history: append a device scalar to a ring (lets a whole multigrid cycle live in one hipGraph)
__global__ void k_append_scalar(const double *__restrict__ src, double *__restrict__ ring, int *__restrict__ count, int cap)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int k = *count;
        if (k < cap) ring[k] = src[0];
        *count = k + 1;
    }
}

// ------------------------------------------------------------------------------------------
// mg_restrict (mg_loops.cpp:30-202) as a coarse-centred gather: coarse = (sum of children in
// ascending fine id) * (1/count); coarse nodes without children keep their value.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_restrict(int64_t nel_coarse, int64_t stride_coarse, int64_t stride_fine, const int32_t *__restrict__ child_ptr,
           const int32_t *__restrict__ child, const int4 *__restrict__ child4, const double *__restrict__ fine_q,
           double *__restrict__ coarse_q, const double *__restrict__ cbrt_vol,
           double *__restrict__ partial_min /* nullptr, or: look ahead, see below */,
           SumTask rms /* .partial != nullptr: the last workgroup also adds up the fine level's per-tile sums of squares
                          (calc_rms of the sweep just finished) — the launch that would do only that is saved */)
{
    // (coarse tiles next to each other read children next to each other: one XCD's L2 per contiguous range of them)
    const int64_t c = xcd_contiguous_block(blockIdx.x, gridDim.x) * int64_t(kBlock) + threadIdx.x;
    double sf = __longlong_as_double(0x7FF0000000000000LL);          // +inf
    if (c < nel_coarse) {
        // The first four children come from a fixed-stride table (-1 padded): their ids need no
        // pointer look-up and their 20 loads are all in flight together; the sum keeps the reference's
        // order (ascending original fine id, mg_loops.cpp:119-142).
        const int4 k4 = child4[c];
        const int32_t b = child_ptr[c], e = child_ptr[c + 1];
        const int32_t n = e - b;
        const int64_t j0 = k4.x < 0 ? 0 : k4.x, j1 = k4.y < 0 ? 0 : k4.y, j2 = k4.z < 0 ? 0 : k4.z, j3 = k4.w < 0 ? 0 : k4.w;
        const int64_t sfn = stride_fine;
        const double a0 = fine_q[j0], a1 = fine_q[sfn + j0], a2 = fine_q[2 * sfn + j0], a3 = fine_q[3 * sfn + j0], a4 = fine_q[4 * sfn + j0];
        const double b0 = fine_q[j1], b1 = fine_q[sfn + j1], b2 = fine_q[2 * sfn + j1], b3 = fine_q[3 * sfn + j1], b4 = fine_q[4 * sfn + j1];
        const double c0 = fine_q[j2], c1 = fine_q[sfn + j2], c2 = fine_q[2 * sfn + j2], c3 = fine_q[3 * sfn + j2], c4 = fine_q[4 * sfn + j2];
        const double d0 = fine_q[j3], d1 = fine_q[sfn + j3], d2 = fine_q[2 * sfn + j3], d3 = fine_q[3 * sfn + j3], d4 = fine_q[4 * sfn + j3];
        double n0, n1, n2, n3, n4;
        if (n == 0) {
            // no children: the coarse node keeps its value (mg_loops.cpp:63-78,174-189)
            if (partial_min) {
                n0 = coarse_q[c]; n1 = coarse_q[stride_coarse + c]; n2 = coarse_q[2 * stride_coarse + c];
                n3 = coarse_q[3 * stride_coarse + c]; n4 = coarse_q[4 * stride_coarse + c];
            }
        } else {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;
            s0 += a0; s1 += a1; s2 += a2; s3 += a3; s4 += a4;
            if (n > 1) { s0 += b0; s1 += b1; s2 += b2; s3 += b3; s4 += b4; }
            if (n > 2) { s0 += c0; s1 += c1; s2 += c2; s3 += c3; s4 += c4; }
            if (n > 3) { s0 += d0; s1 += d1; s2 += d2; s3 += d3; s4 += d4; }
            for (int32_t k = b + 4; k < e; k++) {                     // fifth child onwards (rare)
                const int64_t j = child[k];
                s0 += fine_q[j]; s1 += fine_q[sfn + j]; s2 += fine_q[2 * sfn + j];
                s3 += fine_q[3 * sfn + j]; s4 += fine_q[4 * sfn + j];
            }
            const double average = 1.0 / double(n);
            n0 = s0 * average; n1 = s1 * average; n2 = s2 * average; n3 = s3 * average; n4 = s4 * average;
            store_conserved(coarse_q, stride_coarse, c, n0, n1, n2, n3, n4);
        }
        // The sweep that follows on the coarse level starts with compute_step_factor on exactly these
        // values: leave its first half (the per-workgroup minima) behind.
        if (partial_min) sf = local_step_factor(n0, n1, n2, n3, n4, cbrt_vol[c]);
    }
    if (partial_min) block_min_to(sf, partial_min);
    // (the workgroup dispatched FIRST adds them up: its extra microsecond ends long before the launch's last workgroups do)
    if (rms.partial && blockIdx.x == 0) block_sum_partials(rms);                  // (uniform per workgroup)
}

// ------------------------------------------------------------------------------------------
// prolong_residuals_interpolate_proper (mg_loops.cpp:678-864) as a fine-node gather over the
// same sliced-ELL rows as the flux (one entry per incident internal edge, reference order):
//   wavg = sum_e (w_own*R[p_own] + w_other*R[p_other]) / w_sum   (or R[parent] if coincident)
//   variables += residuals - wavg
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_prolong(int64_t nel, int64_t stride, int64_t stride_coarse, const int32_t *__restrict__ slice_row0,
          const int32_t *__restrict__ rows_int, const double *__restrict__ pro_w, const int32_t *__restrict__ pro_p,
          const int32_t *__restrict__ pro_parent, const double *__restrict__ pro_wsum,
          const double *__restrict__ coarse_residuals, const double *__restrict__ fine_residuals,
          double *__restrict__ fine_q, const double *__restrict__ cbrt_vol,
          double *__restrict__ partial_min /* nullptr, or: look ahead as in k_restrict */)
{
    const int64_t i = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int32_t slice = __builtin_amdgcn_readfirstlane(static_cast<int32_t>(i >> 6));
    if ((int64_t(slice) << 6) >= nel) {                              // a wave past the last node
        if (partial_min) block_min_to(__longlong_as_double(0x7FF0000000000000LL), partial_min);
        return;
    }
    const bool active = i < nel;
    const int64_t ii = active ? i : nel - 1;
    const int32_t row0 = slice_row0[slice];
    const int32_t n_int = rows_int[slice];
    const int32_t parent = pro_parent[ii];
    const int64_t sc = stride_coarse;
    // this node's own state and residual: needed at the end, requested now
    const double q0 = fine_q[ii], q1 = fine_q[stride + ii], q2 = fine_q[2 * stride + ii], q3 = fine_q[3 * stride + ii],
                 q4 = fine_q[4 * stride + ii];
    const double f0 = fine_residuals[ii], f1 = fine_residuals[stride + ii], f2 = fine_residuals[2 * stride + ii],
                 f3 = fine_residuals[3 * stride + ii], f4 = fine_residuals[4 * stride + ii];
    const double ws = pro_wsum[ii];
    // The node's own parent appears in every entry (and, through the reference's b1-for-a1 quirk,
    // as BOTH terms of every entry in which this node is the edge's 'b' end): fetch it once.
    const int64_t own = parent < 0 ? int64_t(~parent) : int64_t(parent);
    const double o0 = coarse_residuals[own], o1 = coarse_residuals[sc + own], o2 = coarse_residuals[2 * sc + own],
                 o3 = coarse_residuals[3 * sc + own], o4 = coarse_residuals[4 * sc + own];
    double r0, r1, r2, r3, r4;
    if (parent < 0) {
        r0 = o0; r1 = o1; r2 = o2; r3 = o3; r4 = o4;
    } else {
        r0 = r1 = r2 = r3 = r4 = 0.0;
        // entries stored [row][w_own | w_other][64 lanes] and [row][64 lanes] (the other end's parent)
        const double *wr = pro_w + (int64_t(row0) << 7) + lane;
        const int32_t *pr = pro_p + (int64_t(row0) << 6) + lane;
        for (int32_t r = 0; r < n_int; r++, wr += 128, pr += 64) {
            const double w_own = wr[0], w_other = wr[64];
            const int32_t p_other = pr[0];
            if (w_own == 0.0 && w_other == 0.0) continue;            // ELL padding
            r0 += w_own * o0; r1 += w_own * o1; r2 += w_own * o2; r3 += w_own * o3; r4 += w_own * o4;
            double x0 = o0, x1 = o1, x2 = o2, x3 = o3, x4 = o4;
            if (p_other != parent) {                                   // the other end's parent: gather
                const int64_t px = p_other;
                x0 = coarse_residuals[px]; x1 = coarse_residuals[sc + px]; x2 = coarse_residuals[2 * sc + px];
                x3 = coarse_residuals[3 * sc + px]; x4 = coarse_residuals[4 * sc + px];
            }
            r0 += w_other * x0; r1 += w_other * x1; r2 += w_other * x2; r3 += w_other * x3; r4 += w_other * x4;
        }
    }
    double sf = __longlong_as_double(0x7FF0000000000000LL);          // +inf
    if (active) {
        const double n0 = q0 + (f0 - r0 / ws);
        const double n1 = q1 + (f1 - r1 / ws);
        const double n2 = q2 + (f2 - r2 / ws);
        const double n3 = q3 + (f3 - r3 / ws);
        const double n4 = q4 + (f4 - r4 / ws);
        store_conserved(fine_q, stride, i, n0, n1, n2, n3, n4);
        if (partial_min) sf = local_step_factor(n0, n1, n2, n3, n4, cbrt_vol[i]);
    }
    if (partial_min) block_min_to(sf, partial_min);
}

// ------------------------------------------------------------------------------------------
// The same prolongation with the coarse residuals served from LDS: one workgroup = one fine tile;
// the distinct coarse nodes the tile refers to (own parents and the other ends' parents, a few
// hundred) are read from HBM once, 40 bytes each, and every entry then finds them in LDS instead
// of gathering five scattered doubles through L1.  Same entries, same order, same arithmetic.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_prolong_tile(int64_t nel, int64_t stride, int64_t stride_coarse, const int32_t *__restrict__ slice_row0,
               const int32_t *__restrict__ rows_int, const double *__restrict__ pro_w,
               const uint16_t *__restrict__ pro_s16, const uint16_t *__restrict__ pro_own16,
               const int32_t *__restrict__ pro_tile_n, const int32_t *__restrict__ pro_tile_ids,
               const int32_t *__restrict__ pro_parent, const double *__restrict__ pro_wsum,
               const double *__restrict__ coarse_residuals, const double *__restrict__ fine_residuals,
               double *__restrict__ fine_q, const double *__restrict__ cbrt_vol, double *__restrict__ partial_min)
{
    __shared__ double cr[kProCap * 5];
    const unsigned t = xcd_contiguous_block(blockIdx.x, gridDim.x);   // neighbouring tiles (they share coarse parents) on one XCD's L2
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int64_t i = int64_t(t) * kTile + tid;
    const bool active = i < nel;
    const int64_t ii = active ? i : nel - 1;
    const int32_t slice = __builtin_amdgcn_readfirstlane(static_cast<int32_t>(i >> 6));
    const int64_t sc = stride_coarse;

    // The tile's coarse residuals are staged behind a dependent chain (ids -> residuals by id): the id of this thread's first staged
    // node goes out FIRST (a fixed-stride table, -1 padded: the address needs only the tile number), everything that depends on
    // nothing — the node's own state, residual, weights — behind it, and the gather last, so that the chain's two round trips are
    // not followed by a third (round 4: the staging loop used to stand in front of every other load).
    const int32_t *ids = pro_tile_ids + int64_t(t) * kProCap;
    const int32_t c_first = ids[tid];
    const int32_t n_ids = pro_tile_n[t];
    const bool wave_live = (int64_t(slice) << 6) < nel;
    const int32_t row0 = wave_live ? slice_row0[slice] : 0;
    const int32_t n_int = wave_live ? rows_int[slice] : 0;
    const int32_t parent = pro_parent[ii];
    const uint32_t own_slot = pro_own16[ii];
    const double q0 = fine_q[ii], q1 = fine_q[stride + ii], q2 = fine_q[2 * stride + ii], q3 = fine_q[3 * stride + ii],
                 q4 = fine_q[4 * stride + ii];
    const double f0 = fine_residuals[ii], f1 = fine_residuals[stride + ii], f2 = fine_residuals[2 * stride + ii],
                 f3 = fine_residuals[3 * stride + ii], f4 = fine_residuals[4 * stride + ii];
    const double ws = pro_wsum[ii];
    // the first row pair's entries go out before the staging barrier too (they do not depend on the staged residuals)
    const double *wr = pro_w + (int64_t(row0) << 7) + lane;
    const uint16_t *sr = pro_s16 + (int64_t(row0) << 6) + lane;
    double wa0 = wr[0], wb0 = wr[64], wa1 = wr[128], wb1 = wr[192];
    uint32_t sl0 = sr[0], sl1 = sr[64];
    {
        // (a thread without a staged node of its own gathers node 0 and stores nothing: the load is never conditional)
        const int64_t c = c_first >= 0 ? c_first : 0;
        const double d0 = coarse_residuals[c], d1 = coarse_residuals[sc + c], d2 = coarse_residuals[2 * sc + c],
                     d3 = coarse_residuals[3 * sc + c], d4 = coarse_residuals[4 * sc + c];
        if (tid < n_ids) { double *d = cr + tid * 5; d[0] = d0; d[1] = d1; d[2] = d2; d[3] = d3; d[4] = d4; }
        for (int32_t k = tid + kBlock; k < n_ids; k += kBlock) {        // (a tile that refers to more than 256 coarse nodes: rare)
            const int64_t c2 = ids[k];
            double *d = cr + k * 5;
            d[0] = coarse_residuals[c2]; d[1] = coarse_residuals[sc + c2]; d[2] = coarse_residuals[2 * sc + c2];
            d[3] = coarse_residuals[3 * sc + c2]; d[4] = coarse_residuals[4 * sc + c2];
        }
    }
    __syncthreads();

    const double *od = cr + own_slot * 5;
    const double o0 = od[0], o1 = od[1], o2 = od[2], o3 = od[3], o4 = od[4];
    double r0, r1, r2, r3, r4;
    if (parent < 0) {
        r0 = o0; r1 = o1; r2 = o2; r3 = o3; r4 = o4;
    } else {
        r0 = r1 = r2 = r3 = r4 = 0.0;
        // entries two rows ahead of their use (the arrays end in padding rows, so the reads past a
        // slice's last row are in bounds and nothing is conditional)
#define MGCFD_PRO_PAIR(R)                                                                                      \
        do {                                                                                                   \
            const bool v0 = wa0 != 0.0 || wb0 != 0.0;                                                          \
            const bool v1 = ((R) + 1 < n_int) && (wa1 != 0.0 || wb1 != 0.0);                                   \
            const double *xd0 = cr + (v0 ? sl0 : own_slot) * 5, *xd1 = cr + (v1 ? sl1 : own_slot) * 5;         \
            const double x00 = xd0[0], x01 = xd0[1], x02 = xd0[2], x03 = xd0[3], x04 = xd0[4];                 \
            const double x10 = xd1[0], x11 = xd1[1], x12 = xd1[2], x13 = xd1[3], x14 = xd1[4];                 \
            if (v0) {                                                                                          \
                r0 += wa0 * o0; r1 += wa0 * o1; r2 += wa0 * o2; r3 += wa0 * o3; r4 += wa0 * o4;                \
                r0 += wb0 * x00; r1 += wb0 * x01; r2 += wb0 * x02; r3 += wb0 * x03; r4 += wb0 * x04;           \
            }                                                                                                  \
            if (v1) {                                                                                          \
                r0 += wa1 * o0; r1 += wa1 * o1; r2 += wa1 * o2; r3 += wa1 * o3; r4 += wa1 * o4;                \
                r0 += wb1 * x10; r1 += wb1 * x11; r2 += wb1 * x12; r3 += wb1 * x13; r4 += wb1 * x14;           \
            }                                                                                                  \
        } while (0)
        // every pair but the last prefetches the pair after it (ELL padding carries zero weights)
        int32_t r = 0;
        for (; r + 2 < n_int; r += 2, wr += 256, sr += 128) {
            const double wa2 = wr[256], wb2 = wr[320], wa3 = wr[384], wb3 = wr[448];
            const uint32_t sl2 = sr[128], sl3 = sr[192];
            MGCFD_PRO_PAIR(r);
            wa0 = wa2; wb0 = wb2; wa1 = wa3; wb1 = wb3; sl0 = sl2; sl1 = sl3;
        }
        if (r < n_int) MGCFD_PRO_PAIR(r);
#undef MGCFD_PRO_PAIR
    }
    double sf = __longlong_as_double(0x7FF0000000000000LL);          // +inf
    if (active) {
        const double n0 = q0 + (f0 - r0 / ws);
        const double n1 = q1 + (f1 - r1 / ws);
        const double n2 = q2 + (f2 - r2 / ws);
        const double n3 = q3 + (f3 - r3 / ws);
        const double n4 = q4 + (f4 - r4 / ws);
        store_conserved(fine_q, stride, i, n0, n1, n2, n3, n4);
        if (partial_min) sf = local_step_factor(n0, n1, n2, n3, n4, cbrt_vol[i]);
    }
    if (partial_min) block_min_to(sf, partial_min);
}

// ==========================================================================================
// launchers
// ==========================================================================================
static inline unsigned grid_for(int64_t n) { return static_cast<unsigned>((n + kBlock - 1) / kBlock); }

void launch_init_variables(hipStream_t st, int64_t stride, const FarField &ff, double *q)
{ hipLaunchKernelGGL(k_init_variables, dim3(grid_for(stride)), dim3(kBlock), 0, st, stride, ff, q); }

// partial_min must hold grid_for(nel) doubles
void launch_step_factor_local(hipStream_t st, int64_t nel, int64_t stride, const double *q, const double *cbrt_vol,
                              double *sf, double *partial_min, double *old_variables)
{ hipLaunchKernelGGL(k_step_factor_local, dim3(grid_for(nel)), dim3(kBlock), 0, st, nel, stride, q, cbrt_vol, sf, partial_min, old_variables); }

void launch_min_reduce(hipStream_t st, int64_t nel, const double *partial_min, double *out)
{ hipLaunchKernelGGL(k_min_reduce, dim3(1), dim3(kBlock), 0, st, partial_min, int(grid_for(nel)), out); }

void launch_step_factor_apply(hipStream_t st, int64_t nel, const double *min_dt_scalar,
                              const double *volumes, double *sf)
{ hipLaunchKernelGGL(k_step_factor_apply, dim3(grid_for(nel)), dim3(kBlock), 0, st, nel, min_dt_scalar, volumes, sf); }

void launch_step_factor_legacy(hipStream_t st, int64_t nel, int64_t stride, const double *q, const double *volumes, double *sf,
                               double *old_variables)
{ hipLaunchKernelGGL(k_step_factor_legacy, dim3(grid_for(nel)), dim3(kBlock), 0, st, nel, stride, q, volumes, sf, old_variables); }

void launch_flux(hipStream_t st, const DevicePlan &p, const double *q, const FarField &ff, double *fluxes,
                 int classes, int accumulate, int variant, const FusedStep *fused, const StagePush *push)
{
    const dim3 block(kBlock);
    FusedStep fs{};
    if (fused) fs = *fused;
    const bool part = fused && fs.tile_list;                        // part of the level's tiles (node gather only)
    const int64_t nel_arg = (fused && fs.nel_active > 0) ? fs.nel_active : p.nel;
    const dim3 grid(part ? fs.n_list : p.n_tiles);
    if (part && fs.n_list <= 0) return;

    // 3 tiles of 52.4 KiB LDS fit a CU (preprocess.hpp: kTileCap) => at least 3 waves per SIMD wanted (168 registers).  The two long-row instantiations that
    // do not fit them — the kernel-granular '+=' launch and the split sweep's absorbed first stage, both off the sweep path —
    // are built for 2 waves per SIMD instead of spilling 20-36 bytes per lane.
#define MGCFD_TILE_LAUNCH_T(WMODE, FUSE, ACC, ROLE, TAIL)                                                      \
    hipLaunchKernelGGL((k_flux_tile<((TAIL) && ((ACC) || (ROLE) == 5)) ? 2 : 3, WMODE, FUSE, ACC, ROLE, TAIL>), grid, block, 0, st, q, p.tile_halo,     \
                       uint32_t(grid.x), p.pad_row, p.stride, nel_arg, p.slice_row0, p.rows_int, p.rows_bnd, \
                       p.nbr16, p.w, p.tile_ovf_ptr, p.tile_ovf, ff, fluxes, classes, fs, p.tail, p.gat16,     \
                       p.te_chunk_ptr, p.te_w3, StagePush{})
    // levels with long rows (tetrahedral meshes, hubs) run the instantiation that hands them to the workgroup
    const bool tail = p.has_tail && (classes & 1);
    // a stage that sends its own message (StagePush): fused stages over a tile list, roles 0-4, k streamed or recomputed
    if (push && fused && part && !fs.vin_flux) {
        const int role_p = fs.partial_min ? 0 : (fs.next_partial_min ? 3 : (fs.next_legacy_sf ? 4 : (fs.residuals ? 2 : 1)));
#define MGCFD_PUSH_LAUNCH_T(WMODE, ROLE, TAIL)                                                                  \
    hipLaunchKernelGGL((k_flux_tile<3, WMODE, true, false, ROLE, TAIL, true>), grid, block, 0, st, q, p.tile_halo,     \
                       uint32_t(grid.x), p.pad_row, p.stride, nel_arg, p.slice_row0, p.rows_int, p.rows_bnd, \
                       p.nbr16, p.w, p.tile_ovf_ptr, p.tile_ovf, ff, fluxes, classes, fs, p.tail, p.gat16,     \
                       p.te_chunk_ptr, p.te_w3, *push)
#define MGCFD_PUSH_LAUNCH_R(WMODE, TAIL)                                                                        \
    do {                                                                                                       \
        if (role_p == 0) MGCFD_PUSH_LAUNCH_T(WMODE, 0, TAIL);                                                  \
        else if (role_p == 2) MGCFD_PUSH_LAUNCH_T(WMODE, 2, TAIL);                                             \
        else if (role_p == 3) MGCFD_PUSH_LAUNCH_T(WMODE, 3, TAIL);                                             \
        else if (role_p == 4) MGCFD_PUSH_LAUNCH_T(WMODE, 4, TAIL);                                             \
        else MGCFD_PUSH_LAUNCH_T(WMODE, 1, TAIL);                                                              \
    } while (0)
        if (role_p == 0 && q != fs.old_variables) throw std::logic_error("a first stage whose input is not the sweep's start state");
        const bool k_streamed = (variant & 1) == 0;
        if (tail) { if (k_streamed) MGCFD_PUSH_LAUNCH_R(1, true); else MGCFD_PUSH_LAUNCH_R(0, true); }
        else { if (k_streamed) MGCFD_PUSH_LAUNCH_R(1, false); else MGCFD_PUSH_LAUNCH_R(0, false); }
#undef MGCFD_PUSH_LAUNCH_R
#undef MGCFD_PUSH_LAUNCH_T
        return;
    }
    // variant bit 4 (16): indexed weights — every edge's weights once per tile (needs the tile edge lists: p.edge_once)
    const bool indexed = (variant & 16) && p.edge_once && p.te_w3 && !tail;
#define MGCFD_TILE_LAUNCH_R(LOADK, FUSE, ACC, ROLE)                                                            \
    do {                                                                                                       \
        if (tail) MGCFD_TILE_LAUNCH_T(LOADK ? 1 : 0, FUSE, ACC, ROLE, true);                                   \
        else if (indexed) MGCFD_TILE_LAUNCH_T(2, FUSE, ACC, ROLE, false);                                      \
        else MGCFD_TILE_LAUNCH_T(LOADK ? 1 : 0, FUSE, ACC, ROLE, false);                                       \
    } while (0)
    // fused stages: the role decides which optional paths exist in the launched kernel
    const int role = !fused ? 1 : (fs.vin_flux ? 5 : fs.partial_min ? 0 : (fs.next_partial_min ? 3 : (fs.next_legacy_sf ? 4 : (fs.residuals ? 2 : 1))));
    // (role 0 takes the sweep's start state from its own record instead of loading old_variables)
    if (role == 0 && q != fs.old_variables) throw std::logic_error("a first stage whose input is not the sweep's start state");
#define MGCFD_TILE_LAUNCH(LOADK, FUSE, ACC)                                                                    \
    do {                                                                                                       \
        if (role == 0) MGCFD_TILE_LAUNCH_R(LOADK, FUSE, ACC, 0);                                               \
        else if (role == 2) MGCFD_TILE_LAUNCH_R(LOADK, FUSE, ACC, 2);                                          \
        else if (role == 3) MGCFD_TILE_LAUNCH_R(LOADK, FUSE, ACC, 3);                                          \
        else if (role == 4) MGCFD_TILE_LAUNCH_R(LOADK, FUSE, ACC, 4);                                          \
        else if (role == 5) MGCFD_TILE_LAUNCH_R(LOADK, FUSE, ACC, 5);                                          \
        else MGCFD_TILE_LAUNCH_R(LOADK, FUSE, ACC, 1);                                                         \
    } while (0)
    // variant bit 2: the two-phase design point (never for the fused stages: they keep the flux in registers)
    if ((variant & 4) && !fused && p.edge_flux) {
        if (classes & 1)
            hipLaunchKernelGGL(k_fission_edge_flux, dim3(grid_for(p.n_edges)), block, 0, st, p.n_edges, p.n_edges_pad, p.stride, q,
                               p.fe_ab, p.fe_w, p.edge_flux);
        if (accumulate)
            hipLaunchKernelGGL(k_fission_node_sum<true>, grid, block, 0, st, p.nel, p.stride, p.n_edges_pad, q, p.slice_row0,
                               p.rows_int, p.rows_bnd, p.row_edge, p.nbr16, p.w, p.edge_flux, ff, fluxes, classes);
        else
            hipLaunchKernelGGL(k_fission_node_sum<false>, grid, block, 0, st, p.nel, p.stride, p.n_edges_pad, q, p.slice_row0,
                               p.rows_int, p.rows_bnd, p.row_edge, p.nbr16, p.w, p.edge_flux, ff, fluxes, classes);
        return;
    }
#ifdef MGCFD_ORDER_FREE
    // variant bit 6 (64), this namespace only: order-free accumulation over the half-row plan (k_flux_free)
    // (not for a launch that must leave the ghost slots alone — fs.nel_active, a partitioned level in direct mode: this kernel
    //  writes every node of the level)
    if ((variant & 64) && p.free_rows && (classes & 1) && !(fused && fs.vin_flux) && !part && !(fused && fs.nel_active > 0)) {
#define MGCFD_FREE_LAUNCH_L(FUSE, ACC, CAP, LONG)                                                               \
    hipLaunchKernelGGL((k_flux_free<FUSE, ACC, CAP, LONG>), grid, block, 0, st, q, p.tile_halo, uint32_t(p.n_tiles), \
                       p.hr_pad_row, p.stride, p.nel, p.hr_row0, p.hr_code, p.hr_w, p.slice_row0,               \
                       p.rows_int, p.rows_bnd, p.nbr16, p.w, ff, fluxes, classes, fs)
#define MGCFD_FREE_LAUNCH_C(FUSE, ACC, CAP)                                                                     \
    do { if (p.hr_max_rows > kHalfMaxRows) MGCFD_FREE_LAUNCH_L(FUSE, ACC, CAP, true); else MGCFD_FREE_LAUNCH_L(FUSE, ACC, CAP, false); } while (0)
        // (a level with halos beyond the shared table: the kernel's own table of kFreeHaloStride ids per tile, 768 LDS images)
#define MGCFD_FREE_LAUNCH_W(FUSE, ACC)                                                                          \
    hipLaunchKernelGGL((k_flux_free<FUSE, ACC, kTile + kFreeHaloStride, true, true>), grid, block, 0, st, q, p.free_halo, uint32_t(p.n_tiles), \
                       p.hr_pad_row, p.stride, p.nel, p.hr_row0, p.hr_code, p.hr_w, p.slice_row0,               \
                       p.rows_int, p.rows_bnd, p.nbr16, p.w, ff, fluxes, classes, fs)
        // (four workgroups per CU where every tile's halo fits the smaller LDS image, three otherwise; MGCFD_FREE_WG3=1: always three, for A/B)
        static const bool wg3 = std::getenv("MGCFD_FREE_WG3") && std::atoi(std::getenv("MGCFD_FREE_WG3")) != 0;
#define MGCFD_FREE_LAUNCH(FUSE, ACC)                                                                            \
    do { if (p.free_wide) MGCFD_FREE_LAUNCH_W(FUSE, ACC);                                                       \
         else if (p.halo_max <= kFreeCap4 - kTile && !wg3) MGCFD_FREE_LAUNCH_C(FUSE, ACC, kFreeCap4); else MGCFD_FREE_LAUNCH_C(FUSE, ACC, kTileCap); } while (0)
        // fused stages of levels on the fast path (halos within the smaller LDS image, at most five half rows per lane): the
        // role-specialised instantiations; fvcorr's look-ahead and every other configuration: the generic epilogue
        const int role_f = !fused ? -1 : (fs.next_legacy_sf ? -1 : (fs.partial_min ? 0 : (fs.next_partial_min ? 3 : ((fs.residuals || fs.sumsq_partial) ? 2 : 1))));
        static const bool no_roles = std::getenv("MGCFD_FREE_NO_ROLES") && std::atoi(std::getenv("MGCFD_FREE_NO_ROLES")) != 0;   // (A/B)
        const bool fast_path = !p.free_wide && p.halo_max <= kFreeCap4 - kTile && !wg3 && p.hr_max_rows <= kHalfMaxRows;
#define MGCFD_FREE_LAUNCH_ROLE(ROLE)                                                                            \
    hipLaunchKernelGGL((k_flux_free<true, false, kFreeCap4, false, false, ROLE>), grid, block, 0, st, q, p.tile_halo, uint32_t(p.n_tiles), \
                       p.hr_pad_row, p.stride, p.nel, p.hr_row0, p.hr_code, p.hr_w, p.slice_row0,               \
                       p.rows_int, p.rows_bnd, p.nbr16, p.w, ff, fluxes, classes, fs)
        if (fused && role_f >= 0 && fast_path && !no_roles && !(role_f == 0 && q != fs.old_variables)) {
            if (role_f == 0) MGCFD_FREE_LAUNCH_ROLE(0);
            else if (role_f == 2) MGCFD_FREE_LAUNCH_ROLE(2);
            else if (role_f == 3) MGCFD_FREE_LAUNCH_ROLE(3);
            else MGCFD_FREE_LAUNCH_ROLE(1);
        }
        else if (fused) MGCFD_FREE_LAUNCH(true, false);
        else if (accumulate) MGCFD_FREE_LAUNCH(false, true);
        else MGCFD_FREE_LAUNCH(false, false);
#undef MGCFD_FREE_LAUNCH_ROLE
#undef MGCFD_FREE_LAUNCH_C
#undef MGCFD_FREE_LAUNCH_L
#undef MGCFD_FREE_LAUNCH_W
#undef MGCFD_FREE_LAUNCH
        return;
    }
#endif
    // variant bit 5 (32): half rows — every edge evaluated once per tile by one of its end points (k_flux_half); the
    // split sweep's absorbed first stage (role 5) stays with the node gather
    if ((variant & 32) && p.half && (classes & 1) && !(fused && fs.vin_flux) && !part) {
#define MGCFD_HALF_LAUNCH(FUSE, ACC)                                                                            \
    hipLaunchKernelGGL((k_flux_half<FUSE, ACC>), grid, block, 0, st, q, p.tile_halo, uint32_t(p.n_tiles),       \
                       p.hr_pad_row, p.stride, p.nel, p.hr_row0, p.hr_code, p.hr_w, p.hg16, p.slice_row0,       \
                       p.rows_int, p.rows_bnd, p.pad_row, p.nbr16, p.w, ff, fluxes, classes, fs)
        if (fused) MGCFD_HALF_LAUNCH(true, false);
        else if (accumulate) MGCFD_HALF_LAUNCH(false, true);
        else MGCFD_HALF_LAUNCH(false, false);
#undef MGCFD_HALF_LAUNCH
        return;
    }
    const bool loadk = (variant & 1) == 0;      // odd variants recompute k = -|e|*s*0.5 from the weights
    // variants 2, 3: every edge evaluated once per tile (needs the internal class and a level whose
    // tiles fit the edge-once limits; otherwise the node gather below)
    if ((variant & 2) && p.edge_once && (classes & 1) && !part) {
#define MGCFD_EO_LAUNCH(LOADK, FUSE, ACC)                                                                      \
    hipLaunchKernelGGL((k_flux_edge_once<LOADK, FUSE, ACC>), grid, block, 0, st, q, p.tile_halo,               \
                       uint32_t(p.n_tiles), p.pad_row, p.stride, p.nel, p.te_chunk_ptr, p.te_count,            \
                       p.pad_chunk, classes, p.slice_row0, p.rows_int, p.rows_bnd, p.nbr16, p.w, p.gat16,      \
                       p.te_slots, p.te_w, p.tile_ovf_ptr, p.tile_ovf, ff, fluxes, fs)
        if (fused) {
            if (loadk) MGCFD_EO_LAUNCH(true, true, false); else MGCFD_EO_LAUNCH(false, true, false);
        } else if (accumulate) {
            if (loadk) MGCFD_EO_LAUNCH(true, false, true); else MGCFD_EO_LAUNCH(false, false, true);
        } else {
            if (loadk) MGCFD_EO_LAUNCH(true, false, false); else MGCFD_EO_LAUNCH(false, false, false);
        }
#undef MGCFD_EO_LAUNCH
        return;
    }
    if (fused) {
        if (loadk) MGCFD_TILE_LAUNCH(true, true, false); else MGCFD_TILE_LAUNCH(false, true, false);
    } else if (accumulate) {
        if (loadk) MGCFD_TILE_LAUNCH(true, false, true); else MGCFD_TILE_LAUNCH(false, false, true);
    } else {
        if (loadk) MGCFD_TILE_LAUNCH(true, false, false); else MGCFD_TILE_LAUNCH(false, false, false);
    }
#undef MGCFD_TILE_LAUNCH
#undef MGCFD_TILE_LAUNCH_R
#undef MGCFD_TILE_LAUNCH_T
}

// The practical ceiling of the flux launch's data movement (bench.py: roofline.practical_ceiling_us): a tile-shaped STREAM of
// exactly the algorithmic bytes — workgroup t reads its share of `rd_total` doubles (contiguous, 16 bytes per lane and load,
// eight loads in flight) and writes its share of `wr_total` (write-through, as the flux launch stores) — with the order-free
// kernel's LDS footprint (four workgroups per CU), nothing dependent, nothing computed.  What the chip gives 1,175 workgroups
// that only move the bytes SURVEY.md §8d prices, launch included.
__global__ void __launch_bounds__(kBlock)
k_stream_tiles(const double2 *__restrict__ src, double *__restrict__ dst, int64_t rd_total2, int64_t wr_total, int rd2_per_thread, int wr_per_thread)
{
    __shared__ double lds[40960 / 8 - 64];
    const int64_t r0 = int64_t(blockIdx.x) * rd2_per_thread * kBlock + threadIdx.x;
    double acc = 0.0;
    for (int k = 0; k < rd2_per_thread; k += 8) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int64_t j = r0 + int64_t(k + u) * kBlock;
            v[u] = src[(k + u < rd2_per_thread && j < rd_total2) ? j : int64_t(threadIdx.x)];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u].x + v[u].y;
    }
    if (acc == 1.2345e300) { lds[threadIdx.x] = acc; __syncthreads(); acc = lds[(threadIdx.x + 1) & 255]; }   // (keeps the footprint)
    const int64_t w0 = int64_t(blockIdx.x) * wr_per_thread * kBlock + threadIdx.x;
    for (int k = 0; k < wr_per_thread; k++) {
        const int64_t j = w0 + int64_t(k) * kBlock;
        if (j < wr_total) MGCFD_ST_FLUX(dst + j, acc + double(k));
    }
}

void launch_stream_tiles(hipStream_t st, int n_tiles, const double *src, double *dst, int64_t rd_total, int64_t wr_total)
{
    const int64_t rd2 = rd_total / 2;
    const int rd2_per_thread = int((rd2 + int64_t(n_tiles) * kBlock - 1) / (int64_t(n_tiles) * kBlock));
    const int wr_per_thread = int((wr_total + int64_t(n_tiles) * kBlock - 1) / (int64_t(n_tiles) * kBlock));
    hipLaunchKernelGGL(k_stream_tiles, dim3(n_tiles), dim3(kBlock), 0, st, reinterpret_cast<const double2 *>(src), dst, rd2, wr_total, rd2_per_thread, wr_per_thread);
}

void launch_indirect_rw(hipStream_t st, const DevicePlan &p, const double *q, double *fluxes, int variant)
{
    // through the flux kernel's LDS tiles where the level allows it (no halo node left outside LDS, no long rows);
    // variant bit 3 (8) forces the plain L1 gather
    if (p.lds_complete && !p.has_tail && !(variant & 8)) {
        if ((variant & 1) == 0)
            hipLaunchKernelGGL((k_indirect_rw_tile<true>), dim3(p.n_tiles), dim3(kBlock), 0, st, q, p.tile_halo, uint32_t(p.n_tiles),
                               p.pad_row, p.stride, p.nel, p.slice_row0, p.rows_int, p.nbr16, p.w, fluxes);
        else
            hipLaunchKernelGGL((k_indirect_rw_tile<false>), dim3(p.n_tiles), dim3(kBlock), 0, st, q, p.tile_halo, uint32_t(p.n_tiles),
                               p.pad_row, p.stride, p.nel, p.slice_row0, p.rows_int, p.nbr16, p.w, fluxes);
        return;
    }
    hipLaunchKernelGGL(k_indirect_rw, dim3(p.n_tiles), dim3(kBlock), 0, st, p.nel, p.stride, q,
                       p.slice_row0, p.rows_int, p.nbr, p.w, fluxes);
}

void launch_time_step(hipStream_t st, int64_t nel, int64_t stride, int j, double *sf, double *fluxes,
                      const double *old_variables, double *q, const int32_t *old_of_new, unsigned long long *err, int check,
                      const double *partial_min, int n_partial, const double *volumes, double *residuals, int zero_fluxes)
{
    const double rk_div = double(3 + 1 - j);    // double(RK+1-j), cfd_loops.cpp:243
    hipLaunchKernelGGL(k_time_step, dim3(grid_for(nel)), dim3(kBlock), 0, st, nel, stride, rk_div, sf, fluxes,
                       old_variables, q, old_of_new, err, check, partial_min, n_partial, volumes, residuals,
                       zero_fluxes);
}

void launch_check_invalid(hipStream_t st, int64_t nel, int64_t stride, const double *q, const int32_t *old_of_new,
                          unsigned long long *err)
{ hipLaunchKernelGGL(k_check_invalid, dim3(grid_for(nel)), dim3(kBlock), 0, st, nel, stride, q, old_of_new, err); }

void launch_residual(hipStream_t st, int64_t stride, const double *old_variables, const double *q, double *residuals)
{ hipLaunchKernelGGL(k_residual, dim3(grid_for(stride * 5)), dim3(kBlock), 0, st, stride * 5, old_variables, q, residuals); }

void launch_sumsq(hipStream_t st, int64_t nel, int64_t stride, const double *x, double *partial, int n_partial, double *out,
                  const int32_t *old_of_new, int64_t n_owned)
{
    hipLaunchKernelGGL(k_sumsq, dim3(n_partial), dim3(kBlock), 0, st, nel, stride, x, partial,
                       n_owned < nel ? old_of_new : nullptr, n_owned);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(kBlock), 0, st, SumTask{partial, n_partial, out, nullptr, nullptr, 0});
}

void launch_halo_pack(hipStream_t st, int64_t n, int64_t stride, const int32_t *idx, const double *field, double *msg)
{ if (n > 0) hipLaunchKernelGGL(k_halo_pack, dim3(grid_for(n)), dim3(kBlock), 0, st, n, stride, idx, field, msg); }

void launch_halo_push(hipStream_t st, int64_t n, int64_t stride, const int32_t *idx, const int32_t *target, const double *field, const PushPeers &peers)
{ if (n > 0) hipLaunchKernelGGL(k_halo_push, dim3(grid_for(n)), dim3(kBlock), 0, st, n, stride, idx, target, field, peers); }

void launch_halo_push_flags(hipStream_t st, int64_t n, int64_t stride, const int32_t *idx, const int32_t *target, const double *field,
                            const PushPeers &peers, const PushFlags &flags, unsigned *ticket)
{ if (n > 0) hipLaunchKernelGGL(k_halo_push_flags, dim3(grid_for(n)), dim3(kBlock), 0, st, n, stride, idx, target, field, peers, flags, ticket); }

void launch_flags_wait(hipStream_t st, const unsigned long long *flags, const FlagRows &rows, int slot, unsigned long long value, int *timed_out)
{ if (rows.n > 0) hipLaunchKernelGGL(k_flags_wait, dim3(1), dim3(64), 0, st, flags, rows, slot, value, timed_out); }

void launch_min_publish(hipStream_t st, const double *my_min, const MinPublish &mp)
{ hipLaunchKernelGGL(k_min_publish, dim3(1), dim3(64), 0, st, my_min, mp); }

void launch_halo_unpack(hipStream_t st, int64_t n, int64_t stride, const int32_t *idx, const double *msg, double *field)
{ if (n > 0) hipLaunchKernelGGL(k_halo_unpack, dim3(grid_for(n)), dim3(kBlock), 0, st, n, stride, idx, msg, field); }

void launch_accept_restricted(hipStream_t st, int64_t nel_coarse, int64_t stride_coarse, const int32_t *child_ptr, const double *src, double *coarse_q)
{ hipLaunchKernelGGL(k_accept_restricted, dim3(grid_for(nel_coarse)), dim3(kBlock), 0, st, nel_coarse, stride_coarse, child_ptr, src, coarse_q); }

void launch_min_over_peers(hipStream_t st, const double *const *scalars, int n, double *out)
{ hipLaunchKernelGGL(k_min_over_peers, dim3(1), dim3(64), 0, st, scalars, n, out); }

void launch_sum_partials_append(hipStream_t st, int n, const double *partial, double *out, double *ring, int *count, int cap)
{ hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(kBlock), 0, st, SumTask{partial, n, out, ring, count, cap}); }

void launch_append_scalar(hipStream_t st, const double *src, double *ring, int *count, int cap)
{ hipLaunchKernelGGL(k_append_scalar, dim3(1), dim3(64), 0, st, src, ring, count, cap); }

void launch_restrict(hipStream_t st, int64_t nel_coarse, int64_t stride_coarse, int64_t stride_fine,
                     const int32_t *child_ptr, const int32_t *child, const int32_t *child4, const double *fine_q,
                     double *coarse_q, const double *cbrt_vol, double *partial_min, const SumTask &rms)
{
    hipLaunchKernelGGL(k_restrict, dim3(grid_for(nel_coarse)), dim3(kBlock), 0, st, nel_coarse, stride_coarse,
                       stride_fine, child_ptr, child, reinterpret_cast<const int4 *>(child4), fine_q, coarse_q, cbrt_vol,
                       partial_min, rms);
}

void launch_prolong(hipStream_t st, const DevicePlan &p, int64_t stride_coarse, const double *coarse_residuals,
                    const double *fine_residuals, double *fine_q, const double *cbrt_vol, double *partial_min)
{
    // grid_for(nel) workgroups: the same partition k_step_factor_local's partial minima use
    if (p.pro_tiled) {
        hipLaunchKernelGGL(k_prolong_tile, dim3(grid_for(p.nel)), dim3(kBlock), 0, st, p.nel, p.stride, stride_coarse,
                           p.slice_row0, p.rows_int, p.pro_w, p.pro_s16, p.pro_own16, p.pro_tile_n, p.pro_tile_ids,
                           p.pro_parent, p.pro_wsum, coarse_residuals, fine_residuals, fine_q, cbrt_vol, partial_min);
        return;
    }
    hipLaunchKernelGGL(k_prolong, dim3(grid_for(p.nel)), dim3(kBlock), 0, st, p.nel, p.stride, stride_coarse,
                       p.slice_row0, p.rows_int, p.pro_w, p.pro_p, p.pro_parent, p.pro_wsum, coarse_residuals, fine_residuals, fine_q,
                       cbrt_vol, partial_min);
}

} // namespace MGCFD_KERNEL_NS
} // namespace mgcfd

#if defined(MGCFD_PHASES) && defined(MGCFD_PHASE_EXPORT)
extern "C" void mgcfd_debug_phases(unsigned long long *out, int reset)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(mgcfd::MGCFD_KERNEL_NS::g_phase), sizeof(unsigned long long) * 4096 * 8);
    if (reset) { static unsigned long long z[4096 * 8]; (void)hipMemcpyToSymbol(HIP_SYMBOL(mgcfd::MGCFD_KERNEL_NS::g_phase), z, sizeof(z)); }
}
extern "C" void mgcfd_debug_phase_abs(unsigned long long *out)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(mgcfd::MGCFD_KERNEL_NS::g_phase_abs), sizeof(unsigned long long) * 4096 * 8);
}
#endif
