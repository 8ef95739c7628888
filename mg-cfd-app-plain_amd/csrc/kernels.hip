// kernels.hip — hand-written HIP kernels for gfx950 (MI355X) for the MG-CFD hot path.
//
// Compiled TWICE into two namespaces:
//   -DMGCFD_KERNEL_NS=exact -ffp-contract=off   every fp64 op rounds as in the reference built
//                                               with -ffp-contract=off; together with the
//                                               reference-order gathers this is bit-identical
//   -DMGCFD_KERNEL_NS=fast  -ffp-contract=fast  same code, FMA contraction allowed
//
// Design: there is no dense contraction here, so no MFMA; every loop is a
// node-centred sweep with one lane per node.
//   * flux_gather   replaces the reference's edge loop + scatter-add
//                   (src/Kernels/flux_loops.cpp:133-136, flux_kernel.elemfunc.c) by a per-node
//                   gather over a sliced-ELLPACK incidence list: coalesced 36 B/entry streams,
//                   96-byte neighbour records fetched with 16-byte loads, a sequential per-node
//                   sum in the reference's order, no atomics, no colouring, deterministic.
//   * The division / square-root work (8 div + 5 sqrt per edge in the reference) is hoisted to
//     one per-node "derive" (3 div + 2 sqrt per node) that is fused into the kernels that
//     produce `variables` (step factor, time step); values are identical because the
//     reference recomputes the very same per-node expressions for every incident edge.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_plan.hpp"

#ifndef MGCFD_KERNEL_NS
#error "compile with -DMGCFD_KERNEL_NS=exact|fast"
#endif

namespace mgcfd {
namespace MGCFD_KERNEL_NS {

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

__device__ __forceinline__ void store_nodeq(NodeQ *q, int64_t i, double rho, double mx, double my,
                                            double mz, double en, const Derived &d)
{
    double2 *dst = reinterpret_cast<double2 *>(q + i);
    dst[0] = make_double2(rho, mx);
    dst[1] = make_double2(my, mz);
    dst[2] = make_double2(en, d.vx);
    dst[3] = make_double2(d.vy, d.vz);
    dst[4] = make_double2(d.p, d.speed);
    dst[5] = make_double2(d.c, 0.0);
}

__device__ __forceinline__ NodeQ load_nodeq(const NodeQ *q, int64_t i)
{
    const double2 *src = reinterpret_cast<const double2 *>(q + i);
    const double2 a = src[0], b = src[1], c = src[2], d = src[3], e = src[4], f = src[5];
    NodeQ r;
    r.rho = a.x; r.mx = a.y; r.my = b.x; r.mz = b.y; r.en = c.x; r.vx = c.y;
    r.vy = d.x; r.vz = d.y; r.p = e.x; r.speed = e.y; r.c = f.x; r.pad = 0.0;
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

__device__ __forceinline__ double wave_min(double v)
{
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
    return v;
}

__device__ __forceinline__ double wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

} // namespace

// ------------------------------------------------------------------------------------------
// initialize_variables (cfd_loops.h:44-55) + first derive
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_init_variables(int64_t nel, FarField ff, double *__restrict__ variables, NodeQ *__restrict__ nodeq)
{
    const int64_t i = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    if (i >= nel) return;
    double *v = variables + i * 5;
    v[0] = ff.var[0]; v[1] = ff.var[1]; v[2] = ff.var[2]; v[3] = ff.var[3]; v[4] = ff.var[4];
    const Derived d = derive(ff.var[0], ff.var[1], ff.var[2], ff.var[3], ff.var[4]);
    store_nodeq(nodeq, i, ff.var[0], ff.var[1], ff.var[2], ff.var[3], ff.var[4], d);
}

// variables -> nodeq (after restrict / prolong / set_array changed variables)
__global__ void __launch_bounds__(kBlock)
k_derive(int64_t nel, const double *__restrict__ variables, NodeQ *__restrict__ nodeq)
{
    const int64_t i = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    if (i >= nel) return;
    const double *v = variables + i * 5;
    const double rho = v[0], mx = v[1], my = v[2], mz = v[3], en = v[4];
    store_nodeq(nodeq, i, rho, mx, my, mz, en, derive(rho, mx, my, mz, en));
}

// ------------------------------------------------------------------------------------------
// compute_step_factor, first half (cfd_loops.cpp:98-125): sf = 0.5 * cbrt(vol) / (|v| + c) and
// the minimum over the level.  cbrt(vol) is static and precomputed on the host with the same
// libm the reference would call.  Also refreshes nodeq (same derive).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_step_factor_local(int64_t nel, const double *__restrict__ variables, const double *__restrict__ cbrt_vol,
                    double *__restrict__ step_factors, NodeQ *__restrict__ nodeq,
                    unsigned long long *__restrict__ min_bits)
{
    __shared__ double s_min[kBlock / 64];
    const int64_t i = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    double sf = __longlong_as_double(0x7F7F7F7F7F7F7F7FLL);
    if (i < nel) {
        const double *v = variables + i * 5;
        const double rho = v[0], mx = v[1], my = v[2], mz = v[3], en = v[4];
        const Derived d = derive(rho, mx, my, mz, en);
        store_nodeq(nodeq, i, rho, mx, my, mz, en, d);
        const double dt = cbrt_vol[i] / (d.speed + d.c);
        sf = 0.5 * dt;
        step_factors[i] = sf;
    }
    sf = wave_min(sf);
    if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = sf;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = s_min[0];
        for (int w = 1; w < kBlock / 64; w++) m = fmin(m, s_min[w]);
        // positive doubles order like their bit patterns
        atomicMin(min_bits, static_cast<unsigned long long>(__double_as_longlong(m)));
    }
}

// second half (cfd_loops.cpp:146-156): step_factors[i] = min_dt / volumes[i]
__global__ void __launch_bounds__(kBlock)
k_step_factor_apply(int64_t nel, const unsigned long long *__restrict__ min_bits,
                    const double *__restrict__ volumes, double *__restrict__ step_factors)
{
    const int64_t i = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    if (i >= nel) return;
    const double min_dt = __longlong_as_double(static_cast<long long>(*min_bits));
    step_factors[i] = min_dt / volumes[i];
}

// compute_step_factor_legacy (cfd_loops.cpp:37-61), mesh_name = fvcorr only
__global__ void __launch_bounds__(kBlock)
k_step_factor_legacy(int64_t nel, const double *__restrict__ variables, const double *__restrict__ volumes,
                     double *__restrict__ step_factors, NodeQ *__restrict__ nodeq)
{
    const int64_t i = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    if (i >= nel) return;
    const double *v = variables + i * 5;
    const double rho = v[0], mx = v[1], my = v[2], mz = v[3], en = v[4];
    const Derived d = derive(rho, mx, my, mz, en);
    store_nodeq(nodeq, i, rho, mx, my, mz, en, d);
    step_factors[i] = 0.5 / (sqrt(volumes[i]) * (d.speed + d.c));
}

// ------------------------------------------------------------------------------------------
// flux_gather: compute_flux_edge + compute_boundary_flux_edge + compute_wall_flux_edge
// (flux_loops.cpp:10-153) as one node-centred gather.  One lane = one node, one wave = one
// slice of the sliced-ELL plan.  `classes` selects which edge classes take part
// (bit0 internal, bit1 solid wall "-1", bit2 far field "-2"); `accumulate` != 0 starts from the
// value already in `fluxes` (the reference's "+=" when the array is not known to be zero).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_flux_gather(int64_t nel, const NodeQ *__restrict__ nodeq, const int32_t *__restrict__ slice_row0,
              const int32_t *__restrict__ rows_int, const int32_t *__restrict__ rows_bnd,
              const int32_t *__restrict__ nbr, const EdgeW *__restrict__ w, FarField ff,
              double *__restrict__ fluxes, int classes, int accumulate)
{
    const int64_t i = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int32_t slice = __builtin_amdgcn_readfirstlane(static_cast<int32_t>(i >> 6));
    if ((int64_t(slice) << 6) >= nel) return;          // whole wave past the end
    const bool active = i < nel;
    const int64_t ii = active ? i : nel - 1;           // keep addresses valid for idle lanes

    const int32_t row0 = slice_row0[slice];
    const int32_t n_int = rows_int[slice];
    const int32_t n_bnd = rows_bnd[slice];

    const NodeQ me = load_nodeq(nodeq, ii);
    const FluxC fm = flux_contribution(me);

    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0;
    if (accumulate) {
        const double *f = fluxes + ii * 5;
        a0 = f[0]; a1 = f[1]; a2 = f[2]; a3 = f[3]; a4 = f[4];
    }

    if (classes & 1) {
        int64_t e = (int64_t(row0) << 6) + lane;
        for (int32_t r = 0; r < n_int; r++, e += 64) {
            const int32_t code = nbr[e];
            if (code < 0) continue;                                   // ELL padding
            const double2 w01 = reinterpret_cast<const double2 *>(w + e)[0];
            const double2 w23 = reinterpret_cast<const double2 *>(w + e)[1];
            const NodeQ ot = load_nodeq(nodeq, code & kIdMask);
            const FluxC fo = flux_contribution(ot);
            const bool me_is_b = (code & kRoleB) != 0;
            // factor = -|e|*0.2f*0.5 * (speed_a + speed_b + c_a + c_b), left-associated
            // (flux_kernel.elemfunc.c:130-131); only the order of the two sound speeds depends
            // on which end this node is.
            const double c_a = me_is_b ? ot.c : me.c;
            const double c_b = me_is_b ? me.c : ot.c;
            const double factor = w23.y * (((me.speed + ot.speed) + c_a) + c_b);
            const double fx = w01.x, fy = w01.y, fz = w23.x;
            // flux_kernel.elemfunc.c:142-189 seen from this node ("me" - "other"; the b-side
            // sign is folded into fx,fy,fz by the plan)
            a0 += factor * (me.rho - ot.rho) + fx * (me.mx + ot.mx) + fy * (me.my + ot.my) + fz * (me.mz + ot.mz);
            a4 += factor * (me.en - ot.en) + fx * (fm.ex + fo.ex) + fy * (fm.ey + fo.ey) + fz * (fm.ez + fo.ez);
            a1 += factor * (me.mx - ot.mx) + fx * (fm.xx + fo.xx) + fy * (fm.xy + fo.xy) + fz * (fm.xz + fo.xz);
            a2 += factor * (me.my - ot.my) + fx * (fm.xy + fo.xy) + fy * (fm.yy + fo.yy) + fz * (fm.yz + fo.yz);
            a3 += factor * (me.mz - ot.mz) + fx * (fm.xz + fo.xz) + fy * (fm.yz + fo.yz) + fz * (fm.zz + fo.zz);
        }
    }

    if ((classes & 6) && n_bnd > 0) {
        // The reference runs ALL solid-wall faces, then ALL far-field faces; the plan lists a
        // node's faces in that order, so one pass per class keeps its per-node order.
        for (int pass = 0; pass < 2; pass++) {
            const int32_t want = pass == 0 ? kCodeWall : kCodeFar;
            if (!(classes & (pass == 0 ? 2 : 4))) continue;
            int64_t e = ((int64_t(row0) + n_int) << 6) + lane;
            for (int32_t r = 0; r < n_bnd; r++, e += 64) {
                if (nbr[e] != want) continue;
                const double2 w01 = reinterpret_cast<const double2 *>(w + e)[0];
                const double wz = reinterpret_cast<const double *>(w + e)[2];
                const double fx = w01.x, fy = w01.y, fz = wz;
                if (pass == 0) {
                    // flux_boundary_kernel.elemfunc.c:37-64: pressure force only
                    a0 += 0.0;
                    a1 += fx * me.p;
                    a2 += fy * me.p;
                    a3 += fz * me.p;
                    a4 += 0.0;
                } else {
                    // flux_wall_kernel.elemfunc.c:51-88: average with the far-field state
                    a0 += fx * (ff.var[1] + me.mx) + fy * (ff.var[2] + me.my) + fz * (ff.var[3] + me.mz);
                    a4 += fx * (ff.fc_de[0] + fm.ex) + fy * (ff.fc_de[1] + fm.ey) + fz * (ff.fc_de[2] + fm.ez);
                    a1 += fx * (ff.fc_mx[0] + fm.xx) + fy * (ff.fc_mx[1] + fm.xy) + fz * (ff.fc_mx[2] + fm.xz);
                    a2 += fx * (ff.fc_my[0] + fm.xy) + fy * (ff.fc_my[1] + fm.yy) + fz * (ff.fc_my[2] + fm.yz);
                    a3 += fx * (ff.fc_mz[0] + fm.xz) + fy * (ff.fc_mz[1] + fm.yz) + fz * (ff.fc_mz[2] + fm.zz);
                }
            }
        }
    }

    if (active) {
        double *f = fluxes + i * 5;
        f[0] = a0; f[1] = a1; f[2] = a2; f[3] = a3; f[4] = a4;
    }
}

// ------------------------------------------------------------------------------------------
// indirect_rw (indirect_rw_kernel.elemfunc.c:4-94) in gather form: same data movement as
// flux_gather, minimal arithmetic.  a-side gets q_b + (ex, ez, 0, 0, ey); b-side gets q_a.
// The plan stores -0.5*e (a side), so e = -2*w exactly.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_indirect_rw(int64_t nel, const NodeQ *__restrict__ nodeq, const int32_t *__restrict__ slice_row0,
              const int32_t *__restrict__ rows_int, const int32_t *__restrict__ nbr,
              const EdgeW *__restrict__ w, double *__restrict__ fluxes)
{
    const int64_t i = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int32_t slice = __builtin_amdgcn_readfirstlane(static_cast<int32_t>(i >> 6));
    if ((int64_t(slice) << 6) >= nel) return;
    const bool active = i < nel;
    const int64_t ii = active ? i : nel - 1;
    const int32_t row0 = slice_row0[slice];
    const int32_t n_int = rows_int[slice];
    const double *f = fluxes + ii * 5;
    double a0 = f[0], a1 = f[1], a2 = f[2], a3 = f[3], a4 = f[4];
    int64_t e = (int64_t(row0) << 6) + lane;
    for (int32_t r = 0; r < n_int; r++, e += 64) {
        const int32_t code = nbr[e];
        if (code < 0) continue;
        const NodeQ ot = load_nodeq(nodeq, code & kIdMask);
        if (code & kRoleB) {
            a0 += ot.rho; a1 += ot.mx; a2 += ot.my; a3 += ot.mz; a4 += ot.en;
        } else {
            const EdgeW we = w[e];
            a0 += ot.rho + (-2.0 * we.x);
            a1 += ot.mx + (-2.0 * we.z);
            a2 += ot.my;
            a3 += ot.mz;
            a4 += ot.en + (-2.0 * we.y);
        }
    }
    if (active) {
        double *g = fluxes + i * 5;
        g[0] = a0; g[1] = a1; g[2] = a2; g[3] = a3; g[4] = a4;
    }
}

// ------------------------------------------------------------------------------------------
// time_step (cfd_loops.cpp:241-268): variables = old + sf/(RK+1-j) * fluxes ; fluxes = 0.
// Fused: refresh nodeq for the next flux pass and raise the check_for_invalid_variables flag
// (validation.cpp:107-138) — err[0] = code, err[1] = smallest offending ORIGINAL cell id.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_time_step(int64_t nel, double rk_div, const double *__restrict__ step_factors, double *__restrict__ fluxes,
            const double *__restrict__ old_variables, double *__restrict__ variables,
            NodeQ *__restrict__ nodeq, const int32_t *__restrict__ old_of_new,
            unsigned long long *__restrict__ err, int check)
{
    const int64_t i = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    if (i >= nel) return;
    const double factor = step_factors[i] / rk_div;
    const double *o = old_variables + i * 5;
    double *f = fluxes + i * 5;
    double *v = variables + i * 5;
    const double rho = o[0] + factor * f[0];
    const double mx = o[1] + factor * f[1];
    const double my = o[2] + factor * f[2];
    const double mz = o[3] + factor * f[3];
    const double en = o[4] + factor * f[4];
    v[0] = rho; v[1] = mx; v[2] = my; v[3] = mz; v[4] = en;
    f[0] = 0.0; f[1] = 0.0; f[2] = 0.0; f[3] = 0.0; f[4] = 0.0;
    store_nodeq(nodeq, i, rho, mx, my, mz, en, derive(rho, mx, my, mz, en));
    if (check) {
        const bool finite = isfinite(rho) && isfinite(mx) && isfinite(my) && isfinite(mz) && isfinite(en);
        int code = 0;
        if (!finite) code = 1;
        else if (rho < 0.0) code = 2;
        else if (en < 0.0) code = 3;
        if (code) {
            // The reference stops at the first bad cell in original order; keep the smallest
            // original id and its code packed as (id << 8) | code.
            const unsigned long long key = (static_cast<unsigned long long>(old_of_new[i]) << 8) | unsigned(code);
            atomicMin(err, key);
        }
    }
}

// check_for_invalid_variables as a standalone sweep
__global__ void __launch_bounds__(kBlock)
k_check_invalid(int64_t nel, const double *__restrict__ variables, const int32_t *__restrict__ old_of_new,
                unsigned long long *__restrict__ err)
{
    const int64_t i = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    if (i >= nel) return;
    const double *v = variables + i * 5;
    const bool finite = isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2]) && isfinite(v[3]) && isfinite(v[4]);
    int code = 0;
    if (!finite) code = 1;
    else if (v[0] < 0.0) code = 2;
    else if (v[4] < 0.0) code = 3;
    if (code) atomicMin(err, (static_cast<unsigned long long>(old_of_new[i]) << 8) | unsigned(code));
}

// residual (validation.cpp:77-89), flat over nel*5 values
__global__ void __launch_bounds__(kBlock)
k_residual(int64_t n, const double *__restrict__ old_variables, const double *__restrict__ variables,
           double *__restrict__ residuals)
{
    const int64_t k = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    if (k < n) residuals[k] = variables[k] - old_variables[k];
}

// sum of squares for calc_rms (validation.cpp:91-105).  Tree order differs from the
// reference's serial sum; the value is only ever printed with %.3e.
__global__ void __launch_bounds__(kBlock)
k_sumsq(int64_t n, const double *__restrict__ x, double *__restrict__ partial)
{
    __shared__ double s[kBlock / 64];
    double acc = 0.0;
    for (int64_t k = blockIdx.x * int64_t(kBlock) + threadIdx.x; k < n; k += int64_t(gridDim.x) * kBlock)
        acc += x[k] * x[k];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kBlock / 64; w++) t += s[w];
        partial[blockIdx.x] = t;
    }
}

__global__ void __launch_bounds__(kBlock)
k_sum_partials(int n, const double *__restrict__ partial, double *__restrict__ out)
{
    __shared__ double s[kBlock / 64];
    double acc = 0.0;
    for (int k = threadIdx.x; k < n; k += kBlock) acc += partial[k];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kBlock / 64; w++) t += s[w];
        out[0] = t;
    }
}

// ------------------------------------------------------------------------------------------
// mg_restrict (mg_loops.cpp:30-202) as a coarse-centred gather: coarse = (sum of children in
// ascending fine id) * (1/count); coarse nodes without children keep their value.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_restrict(int64_t nel_coarse, const int32_t *__restrict__ child_ptr, const int32_t *__restrict__ child,
           const double *__restrict__ fine_variables, double *__restrict__ coarse_variables)
{
    const int64_t c = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    if (c >= nel_coarse) return;
    const int32_t b = child_ptr[c], e = child_ptr[c + 1];
    if (b == e) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;
    for (int32_t k = b; k < e; k++) {
        const double *v = fine_variables + int64_t(child[k]) * 5;
        s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3]; s4 += v[4];
    }
    const double average = 1.0 / double(e - b);
    double *o = coarse_variables + c * 5;
    o[0] = s0 * average; o[1] = s1 * average; o[2] = s2 * average; o[3] = s3 * average; o[4] = s4 * average;
}

// ------------------------------------------------------------------------------------------
// prolong_residuals_interpolate_proper (mg_loops.cpp:678-864) as a fine-node gather over the
// same sliced-ELL rows as the flux (one entry per incident internal edge, reference order):
//   wavg = sum_e (w_own*R[p_own] + w_other*R[p_other]) / w_sum   (or R[parent] if coincident)
//   variables += residuals - wavg
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_prolong(int64_t nel, const int32_t *__restrict__ slice_row0, const int32_t *__restrict__ rows_int,
          const ProlongW *__restrict__ pro, const int32_t *__restrict__ pro_parent,
          const double *__restrict__ pro_wsum, const double *__restrict__ coarse_residuals,
          const double *__restrict__ fine_residuals, double *__restrict__ fine_variables)
{
    const int64_t i = blockIdx.x * int64_t(kBlock) + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int32_t slice = __builtin_amdgcn_readfirstlane(static_cast<int32_t>(i >> 6));
    if ((int64_t(slice) << 6) >= nel) return;
    const bool active = i < nel;
    const int64_t ii = active ? i : nel - 1;
    const int32_t row0 = slice_row0[slice];
    const int32_t n_int = rows_int[slice];
    const int32_t parent = pro_parent[ii];
    double r0, r1, r2, r3, r4;
    if (parent < 0) {
        const double *R = coarse_residuals + int64_t(~parent) * 5;
        r0 = R[0]; r1 = R[1]; r2 = R[2]; r3 = R[3]; r4 = R[4];
    } else {
        r0 = r1 = r2 = r3 = r4 = 0.0;
        int64_t e = (int64_t(row0) << 6) + lane;
        for (int32_t r = 0; r < n_int; r++, e += 64) {
            const ProlongW pw = pro[e];
            if (pw.w_own == 0.0 && pw.w_other == 0.0) continue;      // ELL padding
            const double *Ro = coarse_residuals + int64_t(pw.p_own) * 5;
            const double *Rx = coarse_residuals + int64_t(pw.p_other) * 5;
            r0 += pw.w_own * Ro[0]; r1 += pw.w_own * Ro[1]; r2 += pw.w_own * Ro[2];
            r3 += pw.w_own * Ro[3]; r4 += pw.w_own * Ro[4];
            r0 += pw.w_other * Rx[0]; r1 += pw.w_other * Rx[1]; r2 += pw.w_other * Rx[2];
            r3 += pw.w_other * Rx[3]; r4 += pw.w_other * Rx[4];
        }
    }
    if (!active) return;
    const double ws = pro_wsum[i];
    const double *q = fine_residuals + i * 5;
    double *v = fine_variables + i * 5;
    v[0] += q[0] - r0 / ws;
    v[1] += q[1] - r1 / ws;
    v[2] += q[2] - r2 / ws;
    v[3] += q[3] - r3 / ws;
    v[4] += q[4] - r4 / ws;
}

// ==========================================================================================
// launchers
// ==========================================================================================
static inline unsigned grid_for(int64_t n) { return static_cast<unsigned>((n + kBlock - 1) / kBlock); }

void launch_init_variables(hipStream_t st, int64_t nel, const FarField &ff, double *variables, NodeQ *nodeq)
{ hipLaunchKernelGGL(k_init_variables, dim3(grid_for(nel)), dim3(kBlock), 0, st, nel, ff, variables, nodeq); }

void launch_derive(hipStream_t st, int64_t nel, const double *variables, NodeQ *nodeq)
{ hipLaunchKernelGGL(k_derive, dim3(grid_for(nel)), dim3(kBlock), 0, st, nel, variables, nodeq); }

void launch_step_factor_local(hipStream_t st, int64_t nel, const double *variables, const double *cbrt_vol,
                              double *sf, NodeQ *nodeq, unsigned long long *min_bits)
{ hipLaunchKernelGGL(k_step_factor_local, dim3(grid_for(nel)), dim3(kBlock), 0, st, nel, variables, cbrt_vol, sf, nodeq, min_bits); }

void launch_step_factor_apply(hipStream_t st, int64_t nel, const unsigned long long *min_bits,
                              const double *volumes, double *sf)
{ hipLaunchKernelGGL(k_step_factor_apply, dim3(grid_for(nel)), dim3(kBlock), 0, st, nel, min_bits, volumes, sf); }

void launch_step_factor_legacy(hipStream_t st, int64_t nel, const double *variables, const double *volumes,
                               double *sf, NodeQ *nodeq)
{ hipLaunchKernelGGL(k_step_factor_legacy, dim3(grid_for(nel)), dim3(kBlock), 0, st, nel, variables, volumes, sf, nodeq); }

void launch_flux_gather(hipStream_t st, const DevicePlan &p, const NodeQ *nodeq, const FarField &ff,
                        double *fluxes, int classes, int accumulate)
{
    hipLaunchKernelGGL(k_flux_gather, dim3(grid_for(int64_t(p.n_slices) * 64)), dim3(kBlock), 0, st, p.nel, nodeq,
                       p.slice_row0, p.rows_int, p.rows_bnd, p.nbr, p.w, ff, fluxes, classes, accumulate);
}

void launch_indirect_rw(hipStream_t st, const DevicePlan &p, const NodeQ *nodeq, double *fluxes)
{
    hipLaunchKernelGGL(k_indirect_rw, dim3(grid_for(int64_t(p.n_slices) * 64)), dim3(kBlock), 0, st, p.nel, nodeq,
                       p.slice_row0, p.rows_int, p.nbr, p.w, fluxes);
}

void launch_time_step(hipStream_t st, int64_t nel, int j, const double *sf, double *fluxes, const double *old_variables,
                      double *variables, NodeQ *nodeq, const int32_t *old_of_new, unsigned long long *err, int check)
{
    const double rk_div = double(3 + 1 - j);    // double(RK+1-j), cfd_loops.cpp:243
    hipLaunchKernelGGL(k_time_step, dim3(grid_for(nel)), dim3(kBlock), 0, st, nel, rk_div, sf, fluxes, old_variables,
                       variables, nodeq, old_of_new, err, check);
}

void launch_check_invalid(hipStream_t st, int64_t nel, const double *variables, const int32_t *old_of_new,
                          unsigned long long *err)
{ hipLaunchKernelGGL(k_check_invalid, dim3(grid_for(nel)), dim3(kBlock), 0, st, nel, variables, old_of_new, err); }

void launch_residual(hipStream_t st, int64_t nel, const double *old_variables, const double *variables, double *residuals)
{ hipLaunchKernelGGL(k_residual, dim3(grid_for(nel * 5)), dim3(kBlock), 0, st, nel * 5, old_variables, variables, residuals); }

void launch_sumsq(hipStream_t st, int64_t n, const double *x, double *partial, int n_partial, double *out)
{
    hipLaunchKernelGGL(k_sumsq, dim3(n_partial), dim3(kBlock), 0, st, n, x, partial);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(kBlock), 0, st, n_partial, partial, out);
}

void launch_restrict(hipStream_t st, int64_t nel_coarse, const int32_t *child_ptr, const int32_t *child,
                     const double *fine_variables, double *coarse_variables)
{ hipLaunchKernelGGL(k_restrict, dim3(grid_for(nel_coarse)), dim3(kBlock), 0, st, nel_coarse, child_ptr, child, fine_variables, coarse_variables); }

void launch_prolong(hipStream_t st, const DevicePlan &p, const double *coarse_residuals, const double *fine_residuals,
                    double *fine_variables)
{
    hipLaunchKernelGGL(k_prolong, dim3(grid_for(int64_t(p.n_slices) * 64)), dim3(kBlock), 0, st, p.nel, p.slice_row0,
                       p.rows_int, p.pro, p.pro_parent, p.pro_wsum, coarse_residuals, fine_residuals, fine_variables);
}

} // namespace MGCFD_KERNEL_NS
} // namespace mgcfd
