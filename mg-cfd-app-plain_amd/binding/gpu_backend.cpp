// gpu_backend.cpp — the reference-side binding of libmgcfd_hip.so.
//
// A maintainer of warwick-hpsc/MG-CFD-app-plain adds THIS file to src/Kernels/ and compiles it INSTEAD of
// flux_loops.cpp, cfd_loops.cpp, mg_loops.cpp and indirect_rw_loop.cpp (Makefile:265-277), linking -lmgcfd_hip; main()
// (src/euler3d_cpu_double.cpp), src/Base/*, src/Monitoring/* and validation.cpp stay as they are.  Every function
// below has the reference's own signature (src/Kernels/flux_loops.h:30-43, cfd_loops.h:13-42, mg_loops.h:27-33,
// indirect_rw_loop.h) and forwards to the C ABI entry that replaces it (include/mgcfd.h).
//
// main() is NOT changed, so it keeps owning every array on the host and keeps running its own host loops between the
// calls (copy<double>, residual, calc_rms, check_for_invalid_variables, dump).  This binding therefore hands each
// call's input arrays to the solver and brings its outputs back — correct for any caller, and as slow as the PCIe
// copies make it: it is the functional drop-in at the reference's loop granularity (oracle/build_ref_gpu_backend.sh
// builds it against the real main(); tests/test_gpu_binding.py checks the variables dump byte for byte).  The fast
// path keeps the state resident: mgcfd_run_cycles, as csrc/main.cpp (euler3d_gpu_double) uses it.
//
// The solver itself is created on the first call from the same input files main() read (the globals `conf`,
// `mesh_variant`, `levels`, `level`: src/Base/globals.h:4-16): mgcfd_mesh_load + mgcfd_create_from_mesh read the grid
// as read_grid does and apply adjust_ewt / dampen_ewt exactly as main() applied them to its own copy.
#include <cstdio>
#include <cstdlib>

#include "common.h"         // reference: src/Base/common.h (edge_neighbour, double3, restrict)
#include "globals.h"        // reference: conf, levels, level, current_kernel, mesh_variant
#include "flux_loops.h"
#include "cfd_loops.h"
#include "mg_loops.h"
#include "indirect_rw_loop.h"
#include "timer.h"          // reference: start_timer / stop_timer (src/Monitoring/timer.h)
#include "loop_stats.h"     // reference: record_iters (src/Monitoring/loop_stats.h)

#include "mgcfd.h"

static mgcfd_solver *g_solver = nullptr;

static void fail(const char *what)
{
    std::fprintf(stderr, "\nERROR: libmgcfd_hip: %s: %s\n", what, mgcfd_last_error());
    std::exit(EXIT_FAILURE);
}
#define GPU_CALL(x) do { if ((x) != MGCFD_OK) fail(#x); } while (0)

static mgcfd_solver *solver()
{
    if (g_solver) return g_solver;
    mgcfd_mesh *mesh = nullptr;
    GPU_CALL(mgcfd_mesh_load(conf.input_file, conf.input_file_directory, conf.mesh_duplicate_count, &mesh));
    if (mgcfd_mesh_num_levels(mesh) != levels || mgcfd_mesh_variant(mesh) != mesh_variant) {
        std::fprintf(stderr, "\nERROR: libmgcfd_hip read a different input than main() did\n");
        std::exit(EXIT_FAILURE);
    }
    const char *dev = std::getenv("MGCFD_DEVICE");
    GPU_CALL(mgcfd_create_from_mesh(mesh, dev ? std::atoi(dev) : 0, &g_solver));
    mgcfd_mesh_free(mesh);
    GPU_CALL(mgcfd_set_option(g_solver, MGCFD_OPT_CHECK_INVALID, 0));     // main() runs its own check_for_invalid_variables
    return g_solver;
}

// host array -> solver / solver -> host array (original numbering, the reference's [node][5] layout)
static void put(int l, int which, const double *host) { GPU_CALL(mgcfd_set_array(solver(), l, which, host)); }
static void get(int l, int which, double *host) { GPU_CALL(mgcfd_get_array(solver(), l, which, host)); }

struct timed {                      // what every reference loop does around its body (e.g. flux_loops.cpp:96-112,142-147)
    timed(int kernel, long first, long last)
    {
        current_kernel = kernel;
        #ifdef TIME
        start_timer();
        #endif
        record_iters(first, last);
    }
    ~timed()
    {
        #ifdef TIME
        stop_timer();
        #endif
    }
};

// ---- src/Kernels/cfd_loops.h:13-42 ------------------------------------------------------------------------------
void compute_step_factor(long nel, const double *restrict variables, const double *restrict volumes, double *restrict step_factors)
{
    timed t(COMPUTE_STEP, 0, nel);
    put(level, MGCFD_ARR_VARIABLES, variables);
    GPU_CALL(mgcfd_compute_step_factor(solver(), level));
    get(level, MGCFD_ARR_STEP_FACTORS, step_factors);
}

void compute_step_factor_legacy(long nel, const double *restrict variables, const double *restrict areas, double *restrict step_factors)
{
    compute_step_factor(nel, variables, areas, step_factors);       // the solver picks the variant from mesh_variant, as main() does
}

void time_step(int j, long nel, const double *restrict step_factors, double *restrict fluxes, const double *restrict old_variables,
               double *restrict variables)
{
    timed t(TIME_STEP, 0, nel);
    put(level, MGCFD_ARR_STEP_FACTORS, step_factors);
    put(level, MGCFD_ARR_FLUXES, fluxes);
    put(level, MGCFD_ARR_OLD_VARIABLES, old_variables);
    GPU_CALL(mgcfd_time_step(solver(), level, j));
    get(level, MGCFD_ARR_VARIABLES, variables);
    get(level, MGCFD_ARR_FLUXES, fluxes);                           // time_step leaves them zero (cfd_loops.cpp:262-266)
}

void zero_fluxes(long nel, double *restrict array)
{
    GPU_CALL(mgcfd_zero_fluxes(solver(), level));
    get(level, MGCFD_ARR_FLUXES, array);
}

// ---- src/Kernels/flux_loops.h:9-43 ------------------------------------------------------------------------------
static void flux_class(int (*op)(mgcfd_solver *, int), const double *variables, double *fluxes)
{
    put(level, MGCFD_ARR_VARIABLES, variables);
    put(level, MGCFD_ARR_FLUXES, fluxes);                           // the loops ADD to what is there
    GPU_CALL(op(solver(), level));
    get(level, MGCFD_ARR_FLUXES, fluxes);
}

void compute_flux_edge(long first_edge, long nedges, const edge_neighbour *restrict edges, const double *restrict variables, double *restrict fluxes)
{
    timed t(COMPUTE_FLUX_EDGE, first_edge, first_edge + nedges);
    flux_class(mgcfd_compute_flux_edge, variables, fluxes);
}

void compute_boundary_flux_edge(long first_edge, long nedges, const edge_neighbour *restrict edges, const double *restrict variables, double *restrict fluxes)
{
    current_kernel = COMPUTE_FLUX_EDGE;                             // (the reference neither times nor counts this loop, flux_loops.cpp:10-42)
    flux_class(mgcfd_compute_boundary_flux_edge, variables, fluxes);
}

void compute_wall_flux_edge(long first_edge, long nedges, const edge_neighbour *restrict edges, const double *restrict variables, double *restrict fluxes)
{
    current_kernel = COMPUTE_FLUX_EDGE;                             // (nor this one, flux_loops.cpp:44-76)
    flux_class(mgcfd_compute_wall_flux_edge, variables, fluxes);
}

// ---- src/Kernels/indirect_rw_loop.h ----------------------------------------------------------------------------
void indirect_rw(long first_edge, long nedges, const edge_neighbour *restrict edges, const double *restrict variables, double *restrict fluxes)
{
    timed t(INDIRECT_RW, first_edge, first_edge + nedges);
    flux_class(mgcfd_indirect_rw, variables, fluxes);
}

// ---- src/Kernels/mg_loops.h:27-33 and the one prolongation main() calls ----------------------------------------
// main() has already advanced `level` to the coarse level when it restricts (euler3d_cpu_double.cpp:529-553)
void mg_restrict(double *restrict variables1, double *restrict variables2, long nel2, long *restrict mapping, long *restrict up_scratch, long mgc)
{
    timed t(RESTRICT, 0, mgc);          // the reference counts its three loops (mg_loops.cpp:61,117,172): mgc, mgc, nel2
    record_iters(0, mgc);
    record_iters(0, nel2);
    put(level - 1, MGCFD_ARR_VARIABLES, variables1);
    put(level, MGCFD_ARR_VARIABLES, variables2);                    // coarse nodes without children keep their value
    GPU_CALL(mgcfd_restrict(solver(), level - 1));
    get(level, MGCFD_ARR_VARIABLES, variables2);
}

// ... and stepped back to the fine level when it prolongs (euler3d_cpu_double.cpp:562,671-680)
void prolong_residuals_interpolate_proper(edge_neighbour *restrict edges, long num_edges, double *restrict residuals1, double *restrict residuals2,
                                          double *restrict variables2, long nel2, long *restrict mapping, double3 *restrict coords1,
                                          double3 *restrict coords2)
{
    timed t(PROLONG, 0, num_edges);     // ... and both loops of the prolongation (mg_loops.cpp:730,...): edges, then nodes
    record_iters(0, nel2);
    put(level + 1, MGCFD_ARR_RESIDUALS, residuals1);
    put(level, MGCFD_ARR_RESIDUALS, residuals2);
    put(level, MGCFD_ARR_VARIABLES, variables2);
    GPU_CALL(mgcfd_prolong(solver(), level));
    get(level, MGCFD_ARR_VARIABLES, variables2);
}
