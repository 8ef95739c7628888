// multi_gpu.hpp — euler3d_gpu_double --gpus N (multi_gpu.cpp).
#pragma once

#include <cstdint>
#include <vector>

#include "mgcfd.h"

namespace multi_gpu {

struct Options {
    int gpus = 1;
    int first_device = 0;
    bool share_device = false;        // --gpus-share-device: every rank on `first_device` (a functional rehearsal on a one-GPU box)
    bool fast_math = false;
    bool partition_levels = false;    // --gpus-partition: split EVERY level of a multigrid input over the ranks even when gpus <= levels
};

class Run {
public:
    struct Impl;
    Run(const mgcfd_mesh *mesh, const Options &opt);      // throws std::runtime_error
    ~Run();
    int ranks() const;
    bool partitioned() const;                              // true: the level(s) split over the ranks; false: one multigrid level per rank
    bool partitioned_hierarchy() const;                    // true: every level of a multigrid input split over the ranks (mgcfd_group_cycles)
    int run_cycles(int cycles, double *rms_out);           // MGCFD_OK or MGCFD_ERR_NAN / NEG_*
    void get_level0(int which, int ncols, double *out) const;   // a level-0 array of the WHOLE mesh, original numbering
    int check_invalid(int level, int64_t *bad_cell) const;      // check_for_invalid_variables on `level` of the whole mesh (original cell id)
    void loop_iters(int level, int cycles, int64_t out[MGCFD_NUM_LOOPS]) const;
private:
    Impl *p;
};

} // namespace multi_gpu
