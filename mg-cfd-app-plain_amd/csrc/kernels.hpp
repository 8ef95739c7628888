// kernels.hpp — launchers of kernels.hip, declared once per numeric flavour.
#pragma once

#include "device_plan.hpp"

#define MGCFD_DECLARE_LAUNCHERS(NS)                                                                                  \
    namespace mgcfd { namespace NS {                                                                                 \
    void launch_init_variables(hipStream_t, int64_t stride, const FarField &, double *q);                            \
    void launch_step_factor_local(hipStream_t, int64_t nel, int64_t stride, const double *q, const double *cbrt_vol, \
                                  double *sf, double *partial_min, double *old_variables);                           \
    void launch_min_reduce(hipStream_t, int64_t nel, const double *partial_min, double *out);                        \
    void launch_step_factor_apply(hipStream_t, int64_t nel, const double *min_dt_scalar,                             \
                                  const double *volumes, double *sf);                                                \
    void launch_step_factor_legacy(hipStream_t, int64_t nel, int64_t stride, const double *q, const double *volumes, \
                                   double *sf, double *old_variables);                                               \
    void launch_flux(hipStream_t, const DevicePlan &, const double *q, const FarField &, double *fluxes,             \
                     int classes, int accumulate, int variant, const FusedStep *fused,                               \
                     const StagePush *push = nullptr);                                                               \
    void launch_indirect_rw(hipStream_t, const DevicePlan &, const double *q, double *fluxes, int variant);                       \
    void launch_stream_tiles(hipStream_t, int n_tiles, const double *src, double *dst, int64_t rd_total, int64_t wr_total);      \
    void launch_time_step(hipStream_t, int64_t nel, int64_t stride, int j, double *sf, double *fluxes,               \
                          const double *old_variables, double *q, const int32_t *old_of_new,                         \
                          unsigned long long *err, int check, const double *partial_min, int n_partial,              \
                          const double *volumes, double *residuals, int zero_fluxes);                                \
    void launch_check_invalid(hipStream_t, int64_t nel, int64_t stride, const double *q,                             \
                              const int32_t *old_of_new, unsigned long long *err);                                   \
    void launch_residual(hipStream_t, int64_t stride, const double *old_variables, const double *q,                  \
                         double *residuals);                                                                         \
    void launch_sumsq(hipStream_t, int64_t nel, int64_t stride, const double *x, double *partial, int n_partial,     \
                      double *out, const int32_t *old_of_new, int64_t n_owned);                                      \
    void launch_halo_pack(hipStream_t, int64_t n, int64_t stride, const int32_t *idx, const double *field,           \
                          double *msg);                                                                              \
    void launch_halo_unpack(hipStream_t, int64_t n, int64_t stride, const int32_t *idx, const double *msg,           \
                            double *field);                                                                          \
    void launch_halo_push(hipStream_t, int64_t n, int64_t stride, const int32_t *idx, const int32_t *target,         \
                          const double *field, const PushPeers &peers);                                              \
    void launch_halo_push_flags(hipStream_t, int64_t n, int64_t stride, const int32_t *idx, const int32_t *target,   \
                                const double *field, const PushPeers &peers, const PushFlags &flags,                 \
                                unsigned *ticket);                                                                   \
    void launch_flags_wait(hipStream_t, const unsigned long long *flags, const FlagRows &rows, int slot,             \
                           unsigned long long value, int *timed_out);                                                \
    void launch_min_publish(hipStream_t, const double *my_min, const MinPublish &mp);                                \
    void launch_append_scalar(hipStream_t, const double *src, double *ring, int *count, int cap);                    \
    void launch_min_over_peers(hipStream_t, const double *const *scalars, int n, double *out);                       \
    void launch_accept_restricted(hipStream_t, int64_t nel_coarse, int64_t stride_coarse, const int32_t *child_ptr,  \
                                  const double *src, double *coarse_q);                                              \
    void launch_sum_partials_append(hipStream_t, int n, const double *partial, double *out, double *ring,            \
                                    int *count, int cap);                                                            \
    void launch_restrict(hipStream_t, int64_t nel_coarse, int64_t stride_coarse, int64_t stride_fine,                \
                         const int32_t *child_ptr, const int32_t *child, const int32_t *child4, const double *fine_q, \
                         double *coarse_q,                                                                        \
                         const double *cbrt_vol, double *partial_min, const SumTask &rms);                          \
    void launch_prolong(hipStream_t, const DevicePlan &, int64_t stride_coarse, const double *coarse_residuals,      \
                        const double *fine_residuals, double *fine_q, const double *cbrt_vol,                       \
                        double *partial_min);                                                                        \
    } }

MGCFD_DECLARE_LAUNCHERS(exact)
MGCFD_DECLARE_LAUNCHERS(fast)
