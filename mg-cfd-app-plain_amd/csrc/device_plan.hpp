// device_plan.hpp — plain structs shared by the host solver and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

#include "preprocess.hpp"

namespace mgcfd {

// Node-state fields per level, stored structure-of-arrays with a common padded stride
// (stride = 64 * n_slices): q[f*stride + i], f = 0..4 conserved variables (the reference's
// `variables`), f = 5..10 the quantities the reference derives from them for every incident
// edge (velocity, pressure, |v|, speed of sound; cfd_loops.h:121-148).
constexpr int kNumStateFields = 11;
static_assert(sizeof(EdgeW) == 32, "EdgeW must be 32 bytes");
static_assert(sizeof(ProlongW) == 24, "ProlongW must be 24 bytes");

// ff_variable + ff_flux_contribution_* (src/Base/globals.h:11-15), passed by value.
struct FarField {
    double var[5];
    double fc_mx[3], fc_my[3], fc_mz[3], fc_de[3];
};

// Device pointers of one level's gather plan.
struct DevicePlan {
    int64_t nel = 0;
    int64_t stride = 0;                 // 64 * n_slices
    int32_t n_slices = 0;
    int32_t *slice_row0 = nullptr, *rows_int = nullptr, *rows_bnd = nullptr, *nbr = nullptr;
    double *w = nullptr;                // [row][4 components][64 lanes]
    int32_t n_tiles = 0;
    int32_t *nbr_tile = nullptr, *tile_halo_ptr = nullptr, *tile_halo = nullptr;
    int32_t *old_of_new = nullptr;
    // transfer to/from the next-coarser level
    int32_t *child_ptr = nullptr, *child = nullptr;
    ProlongW *pro = nullptr;
    int32_t *pro_parent = nullptr;
    double *pro_wsum = nullptr;
};

} // namespace mgcfd
