/*
 * mgcfd.h — C ABI of libmgcfd_hip.so: the MI355X (gfx950) implementation of
 * MG-CFD's edge-flux / multigrid hot path.
 *
 * The reference (warwick-hpsc/MG-CFD-app-plain) has no plugin/FFI layer: its
 * hot path is a set of free functions on caller-owned host arrays
 * (src/Kernels/{flux_loops,cfd_loops,mg_loops,validation}.h) driven by main()
 * (src/euler3d_cpu_double.cpp:371-694).  This header exports that same set at
 * the same granularity behind an opaque handle that keeps the mesh and the
 * state resident in HBM; every entry point names the reference function it
 * replaces.  Plain C types only, no exceptions cross the boundary, no exit():
 * every call returns MGCFD_OK or an error code and mgcfd_last_error() explains.
 *
 * Numbering: callers always see the reference's ORIGINAL node and edge
 * numbering; the library renumbers internally and permutes on the way in/out.
 * Threading: one host thread per solver handle (as the reference's main()).
 */
#ifndef MGCFD_H
#define MGCFD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGCFD_NVAR 5   /* src/Base/const.h:29  (rho, rho*u, rho*v, rho*w, rho*E) */
#define MGCFD_RK 3     /* src/Base/const.h:13 */

/* mesh_name codes — src/Base/const.h:40-43 */
#define MGCFD_MESH_FVCORR 0
#define MGCFD_MESH_M6_WING 2
#define MGCFD_MESH_LA_CASCADE 3
#define MGCFD_MESH_ROTOR_37 4

/* error codes */
#define MGCFD_OK 0
#define MGCFD_ERR_ARG 1        /* bad argument / level out of range */
#define MGCFD_ERR_IO 2         /* file missing or malformed */
#define MGCFD_ERR_HIP 3        /* HIP runtime failure (no device, OOM, launch error) */
#define MGCFD_ERR_NAN 4        /* check_for_invalid_variables: NaN/Inf          (validation.cpp:112-121) */
#define MGCFD_ERR_NEG_DENSITY 5 /* ... negative density                         (validation.cpp:124-128) */
#define MGCFD_ERR_NEG_ENERGY 6  /* ... negative density*energy                  (validation.cpp:130-134) */
#define MGCFD_ERR_VALIDATION 7 /* identify_differences found a value out of tolerance (validation.cpp:140-199) */

/* Loop ids: the columns of Times.csv / LoopNumIters.csv in file order
 * (src/Monitoring/timer.cpp:135-144, src/Base/const.h:31-38 names). */
enum { MGCFD_LOOP_FLUX = 0, MGCFD_LOOP_UPDATE, MGCFD_LOOP_COMPUTE_STEP, MGCFD_LOOP_TIME_STEP,
       MGCFD_LOOP_RESTRICT, MGCFD_LOOP_PROLONG, MGCFD_LOOP_INDIRECT_RW, MGCFD_NUM_LOOPS };

/* Per-level arrays a caller can read back / overwrite (reference: the arrays main() owns,
 * src/euler3d_cpu_double.cpp:138-162). */
enum { MGCFD_ARR_VARIABLES = 0, MGCFD_ARR_OLD_VARIABLES, MGCFD_ARR_FLUXES, MGCFD_ARR_RESIDUALS,
       MGCFD_ARR_STEP_FACTORS, MGCFD_ARR_VOLUMES,
       MGCFD_ARR_STAGE /* the state the last mgcfd_sweep_stage wrote (halo messages between the stages of a split sweep) */ };

/* Solver options (mgcfd_set_option) */
enum {
    MGCFD_OPT_EXACT = 0,       /* 1 (default): kernels compiled without FMA contraction and summing in the
                                  reference's order => bit-identical to the reference built with
                                  -ffp-contract=off, and the same bits from run to run.
                                  0: the fast mode — FMA contraction allowed AND, with MGCFD_OPT_FLUX_VARIANT = -1, the
                                  order-free flux kernel wherever a level has its plan (bit 6 below): within 1e-12 relative
                                  of the reference per sweep (north_star allows 1e-10), but NOT reproducible bit for bit
                                  from run to run, nor between a whole level and the same level partitioned (a launch over
                                  part of a level takes the node gather).  MGCFD_OPT_EXACT = 0 with MGCFD_OPT_FLUX_VARIANT = 1
                                  is the deterministic contracted mode (the node gather only). */
    MGCFD_OPT_TIMING = 1,      /* 1: bracket every loop with hipEvents (Times.csv columns; every loop its own launch, as the
                                  reference's -DTIME build brackets them, src/Monitoring/timer.cpp:58-195); 2: only the flux launches
                                  of every 8th sweep; 3: as 2 for every sweep (reads back as 2);
                                  4: per-loop times by ATTRIBUTION — the cycles run the fused stages; every 32nd sweep and every 32nd
                                  restriction / prolongation of a level runs per loop under events (the indirect_rw probe in
                                  every fourth of those, its column extrapolated at the measured rate), one event pair brackets each batch of cycles, and mgcfd_get_loop_times apportions the
                                  batches' GPU time to (level, loop) by the sampled launches' ratios: all seven columns at
                                  ~1.05x the fused cycle time instead of 2.7x (what euler3d_gpu_double runs by default) */
    MGCFD_OPT_INDIRECT_RW = 2, /* 1: also run the indirect_rw probe each RK stage, as the reference's main() does */
    MGCFD_OPT_CHECK_INVALID = 3, /* 1 (default): NaN / negativity check every RK stage (validation.cpp:107-138) */
    MGCFD_OPT_FLUX_VARIANT = 4, /* -1 (default): automatic — 1 (the edge-length factor recomputed: never slower on an
                                   MI355X, faster once a level's rows outgrow the Infinity Cache).  Otherwise a bit set:
                                   bit 0: 0 edge-length factor streamed, 1 recomputed from the weights;
                                   bit 1: 0 (default) node gather (every edge evaluated from both ends),
                                   2 edge-once tiles (every edge evaluated once per tile; levels whose tiles
                                   hold too many edges fall back to the node gather, see
                                   mgcfd_level_has_edge_once);
                                   bit 2 (4): two-phase design point — edge fluxes written to memory, then a node-centred
                                   sum (the reference's FLUX_FISSION idea); kernel-granular and unfused sweeps only.
                                   bit 4 (16): indexed weights — a tile lists every edge's weights once, 24 B, and the
                                   row entries (4 B) point at them; a design point: the second end point's gather
                                   through L1 costs more than the bytes it saves (24 us against 17.5);
                                   bit 5 (32): half rows — every internal edge of a tile evaluated once, by one of its
                                   end points, the flux terms handed to the other through LDS (k_flux_half; levels that
                                   qualify, see mgcfd_level_has_half_rows; others run the node gather).
                                   Every variant above gives bit-identical results.
                                   bit 6 (64), with MGCFD_OPT_EXACT = 0 only (the bit-identical kernels ignore it): ORDER-FREE
                                   accumulation over the half-row plan (k_flux_free) — every internal edge of a tile evaluated
                                   once, the other end's share added to its LDS sum with fp64 LDS atomics, no ordered hand-over.
                                   Sums are associated differently from the reference's (src/Kernels/flux_loops.cpp:133-136 adds
                                   in edge order) and are not reproducible bit for bit from run to run: <= 1e-12 relative per
                                   launch, <= 1e-10 after 25 V-cycles, the reference's -v rule (validation.cpp:159-166) passes;
                                   tests/test_gpu_order_free.py.  Levels without a half-row plan run the node gather.
                                   With MGCFD_OPT_EXACT = 0 the automatic choice (-1) sets this bit where the kernel is the faster
                                   one: for every standalone flux launch of a level that has the plan, for the fused stages of
                                   levels with long rows or tile halos beyond the shared table (tetrahedral levels) and — round 4:
                                   role-specialised stages — of levels on the kernel's fast path (halos of at most 290 nodes, at
                                   most five evaluations per lane: the lattice-like levels of the named meshes). */
    MGCFD_OPT_FUSE_UPDATE = 5, /* 1 (default): mgcfd_smooth / mgcfd_run_cycles run each Runge-Kutta stage as ONE
                                  launch (fluxes + time_step, same operations); 0: one launch per loop */
    MGCFD_OPT_GRAPH = 6,       /* 1: replay each smoothing sweep / multigrid cycle from a captured hipGraph (one host
                                  call instead of 3 / ~24 launches).  0 (default): launch the kernels directly — on
                                  ROCm 7.2 / MI355X the graph costs ~1.7 us per kernel node more than direct launches
                                  (sweep 75 us replayed, 70 us launched), so it only pays when the host cannot keep
                                  the queue full */
    MGCFD_OPT_RANK_SPLIT = 7   /* ranks in different processes, direct stores (mgcfd_rank_ipc_*): 1 (default) a stage runs its
                                  boundary tiles first, sends, then the interior tiles (the message's flight is hidden);
                                  0: all tiles in ONE launch, then the message (two launches less per stage, the flight
                                  exposed); 2: ONE launch that sends its own message — the boundary tiles come first, their
                                  epilogue stores into the neighbours, the last of them raises the flags while the interior
                                  tiles still run.  Which is fastest depends on the flight time: bench.py times all three. */
};

/* Same 40-byte layout as the reference's edge_neighbour (src/Base/definitions.h:83). */
typedef struct { int64_t a, b; double x, y, z; } mgcfd_edge;

/* One multigrid level, as read_grid()/read_mg_connectivity() produce it
 * (src/Base/io.cpp:14-199, src/Base/io_enhanced.cpp:629-650).  All pointers are
 * caller-owned host memory, copied by mgcfd_create(). */
typedef struct {
    int64_t nel;
    int64_t n_edges;                       /* allocated length of edges[] */
    int64_t n_internal, n_boundary, n_wall;
    int64_t internal_start, boundary_start, wall_start;
    const double *volumes;                 /* [nel] */
    const double *coords;                  /* [nel*3] x y z; may be NULL for a single-level fvcorr run */
    const mgcfd_edge *edges;               /* [n_edges]: internal | boundary (a=-1) | wall (a=-2) */
    const int64_t *mg_map;                 /* [mgc] fine -> coarse map to the next level; NULL on the last */
    int64_t mgc;
} mgcfd_level_desc;

typedef struct mgcfd_mesh mgcfd_mesh;      /* host-side multigrid input (files parsed)   */
typedef struct mgcfd_solver mgcfd_solver;  /* device-resident solver                     */

const char *mgcfd_last_error(void);
int mgcfd_abi_version(void);
/* Optional: start bringing up the HIP runtime and `device`'s context on a thread of the library's own and return at once, so
 * that it happens WHILE the caller reads its input files (0.1 s of a drop-in run's start-up); mgcfd_create* waits for it.
 * No reference counterpart (a CPU code has no device to wake).  Errors surface in mgcfd_create*, not here. */
int mgcfd_device_warm_up(int device);

/* ---------------------------------------------------------------------------------
 * File boundary (host only, no GPU needed)
 * --------------------------------------------------------------------------------- */
/* read_input_dat + read_grid + read_mg_connectivity (+ duplicate_mesh when duplicate > 1):
 * src/Base/io_enhanced.cpp:407-579, src/Base/io.cpp:14-199, io_enhanced.cpp:629-650, :89-201.
 * `directory` may be NULL/"" (paths then relative to the cwd, as with no -d). */
int mgcfd_mesh_load(const char *input_dat, const char *directory, int duplicate, mgcfd_mesh **out);
/* The same with flags: MGCFD_MESH_LEGACY_ORDERING sorts every edge class by (a, b, x, y, z) as the
 * reference's -DLEGACY_ORDERING build does (src/Base/io.cpp:183-193, src/Base/common.h:145-157). */
#define MGCFD_MESH_LEGACY_ORDERING 1
int mgcfd_mesh_load_ex(const char *input_dat, const char *directory, int duplicate, int flags, mgcfd_mesh **out);
void mgcfd_mesh_free(mgcfd_mesh *m);
int mgcfd_mesh_num_levels(const mgcfd_mesh *m);
int mgcfd_mesh_variant(const mgcfd_mesh *m);
int mgcfd_mesh_size(const mgcfd_mesh *m);                 /* input.dat "size" x duplicate (euler3d_cpu_double.cpp:259-260) */
/* Borrowed view of level l (valid until mgcfd_mesh_free). */
int mgcfd_mesh_level(const mgcfd_mesh *m, int level, mgcfd_level_desc *out);

/* dump(): "%.17e" x5 per node (src/Base/io.cpp:201-233); generic writer for 1- or 5-column arrays. */
int mgcfd_write_array(const char *path, const double *data, int64_t nel, int ncols);
/* identify_differences(): returns MGCFD_OK or MGCFD_ERR_VALIDATION; *first_bad = flat index or -1. */
int mgcfd_identify_differences(const double *test_values, const double *master_values, int64_t nel,
                               int mesh_variant, int64_t *first_bad);

/* ---------------------------------------------------------------------------------
 * Solver life cycle
 * --------------------------------------------------------------------------------- */
/* Uploads the levels to GPU `device`, applies adjust_ewt/dampen_ewt for the mesh variant
 * (euler3d_cpu_double.cpp:337-352, validation.cpp:28-75), builds the renumbered gather
 * structures, and initialises every level to the far-field state
 * (initialize_far_field_conditions + initialize_variables, cfd_loops.h:44-119;
 * fluxes/residuals zeroed, euler3d_cpu_double.cpp:321-331). */
int mgcfd_create(const mgcfd_level_desc *levels, int nlevels, int mesh_variant, int device,
                 mgcfd_solver **out);
int mgcfd_create_from_mesh(const mgcfd_mesh *m, int device, mgcfd_solver **out);
/* A level PARTITIONED over ranks (multi-GPU "within a level"): every rank passes its own part —
 * its owned nodes numbered first, then ghost copies of the other ranks' nodes its edges touch, and
 * every edge with at least one owned end point (relative order as in the whole mesh, so sums keep
 * the reference's order).  n_owned[l] = number of owned nodes of level l (nodes with id >= n_owned
 * are ghosts: gathered from, never updated; excluded from the RMS). */
int mgcfd_create_partitioned(const mgcfd_level_desc *levels, int nlevels, int mesh_variant, int device,
                             const int64_t *n_owned, mgcfd_solver **out);
/* The same for a partitioned HIERARCHY (every level split over the ranks, mg_map in local numbering: the parent of
 * every local fine node must be a local coarse node).  order_keys[l] (may be NULL) gives, per node of level l, the key by
 * which a coarse node's children are summed in mgcfd_restrict — global ids, so the mean keeps the whole mesh's order.
 * The caller exchanges ghost values where the hierarchy needs them: `variables` after every time_step, after
 * mgcfd_restrict (coarse level) and after mgcfd_prolong (fine level); coarse `residuals` before mgcfd_prolong
 * (mgcfd/partition.py: partition_hierarchy, mgcfd/distributed.py: PartitionedCycle). */
int mgcfd_create_partitioned_mg(const mgcfd_level_desc *levels, int nlevels, int mesh_variant, int device,
                                const int64_t *n_owned, const int64_t *const *order_keys, mgcfd_solver **out);
/* Host only (no device): builds the gather plans the three calls above would build and checks every index the kernels
 * form from them — LDS slots, halo / overflow positions, half-row owners, list entries, children, staged coarse nodes —
 * against the size of what it indexes.  MGCFD_OK, or MGCFD_ERR_ARG with the violations in `report` (may be NULL).
 * n_owned / order_keys as in mgcfd_create_partitioned_mg (NULL: every node owned).  The reference has no counterpart:
 * its loops index the caller's arrays directly (src/Kernels/flux_loops.cpp:133-136). */
int mgcfd_plan_audit(const mgcfd_level_desc *levels, int nlevels, int mesh_variant, const int64_t *n_owned,
                     const int64_t *const *order_keys, char *report, int64_t report_cap);
void mgcfd_destroy(mgcfd_solver *s);
int mgcfd_set_option(mgcfd_solver *s, int option, int value);
/* *yes = 1 when level `level` can run the edge-once flux variant (MGCFD_OPT_FLUX_VARIANT bit 1). */
int mgcfd_level_has_edge_once(const mgcfd_solver *s, int level, int *yes);
/* *yes = 1 when level `level` can run the half-row flux kernel (MGCFD_OPT_FLUX_VARIANT bit 5): no long rows, no halo node
 * left outside LDS, at most 5 edges evaluated per node. */
int mgcfd_level_has_half_rows(const mgcfd_solver *s, int level, int *yes);
/* *yes = 1 when level `level` can run the order-free flux kernel (MGCFD_OPT_FLUX_VARIANT bit 6 with MGCFD_OPT_EXACT = 0): no halo
 * node left outside LDS; any number of edges per node (what a lane's five requested rows do not hold is walked in a loop). */
int mgcfd_level_has_order_free(const mgcfd_solver *s, int level, int *yes);
/* What the tiling of level `level` looks like (the figures MGCFD_VERBOSE=1 prints at creation):
 * out[0] tiles of 256 nodes, out[1] halo nodes of all tiles together, out[2] the largest halo, out[3] halo nodes a
 * tile can stage in LDS, out[4] incidence-row entries that refer to a halo node beyond that (each a gather from
 * HBM), out[5] incidence-row entries of internal edges (two per edge), out[6] padding entries among them,
 * out[7] 1 when the nodes were ordered by coordinate boxes instead of greedy clusters, out[8] internal-edge entries
 * handed to the workgroups' lists instead of the per-node loops (long rows: entries beyond a tile's row limit, and
 * every entry from a node's first out[4]-kind entry on — so with out[8] > 0 none of out[4] is gathered inside a loop),
 * out[9] rows the per-node loops walk, summed over the 64-node slices. */
int mgcfd_level_tiling(const mgcfd_solver *s, int level, int64_t out[10]);
int mgcfd_get_option(const mgcfd_solver *s, int option, int *value);
/* Run all subsequent work of this solver on an existing HIP stream (hipStream_t as void*), e.g.
 * the cuda_stream of a torch.cuda.Stream() made current with torch.cuda.set_stream(): collectives
 * and copies torch enqueues on that stream are then ordered with the solver's kernels.  NULL restores
 * the solver's own stream, which is created non-blocking: it does NOT synchronise with the legacy
 * default stream, whose handle is also NULL — the default stream cannot be shared this way. */
int mgcfd_set_stream(mgcfd_solver *s, void *hip_stream);
int mgcfd_synchronize(mgcfd_solver *s);
int mgcfd_num_levels(const mgcfd_solver *s);
int64_t mgcfd_level_nel(const mgcfd_solver *s, int level);
int64_t mgcfd_level_num_internal_edges(const mgcfd_solver *s, int level);
/* ff_variable[5] + the four ff_flux_contribution_* vectors (17 doubles), globals.h:11-15. */
int mgcfd_get_far_field(const mgcfd_solver *s, double *out17);

/* ---------------------------------------------------------------------------------
 * Kernel-granular operations (asynchronous on the solver's stream)
 * --------------------------------------------------------------------------------- */
/* copy<double>(old_variables, variables)                    src/Base/common.h:100-112 */
int mgcfd_copy_old_variables(mgcfd_solver *s, int level);
/* compute_step_factor / compute_step_factor_legacy (chosen by mesh variant exactly as
 * euler3d_cpu_double.cpp:388-395 does)                       src/Kernels/cfd_loops.cpp:13-157 */
int mgcfd_compute_step_factor(mgcfd_solver *s, int level);
/* compute_flux_edge over the level's internal edges: fluxes += …   flux_loops.cpp:78-153 */
int mgcfd_compute_flux_edge(mgcfd_solver *s, int level);
/* compute_boundary_flux_edge (neighbour code -1)                   flux_loops.cpp:10-42 */
int mgcfd_compute_boundary_flux_edge(mgcfd_solver *s, int level);
/* compute_wall_flux_edge (neighbour code -2, far field)            flux_loops.cpp:44-76 */
int mgcfd_compute_wall_flux_edge(mgcfd_solver *s, int level);
/* The three above in one launch (same per-node summation order as calling them in sequence). */
int mgcfd_compute_fluxes(mgcfd_solver *s, int level);
/* time_step(j, …): variables = old + sf/(RK+1-j)*fluxes; fluxes = 0    cfd_loops.cpp:215-280 */
int mgcfd_time_step(mgcfd_solver *s, int level, int j);
/* zero_fluxes                                                       cfd_loops.cpp:282-305 */
int mgcfd_zero_fluxes(mgcfd_solver *s, int level);
/* indirect_rw over the internal edges (fluxes += …)                 indirect_rw_loop.cpp:11-78 */
int mgcfd_indirect_rw(mgcfd_solver *s, int level);
/* residual(): residuals = variables - old_variables                 validation.cpp:77-89 */
int mgcfd_residual(mgcfd_solver *s, int level);
/* calc_rms(): sqrt(sum(r^2)/nel); synchronises                      validation.cpp:91-105 */
int mgcfd_calc_rms(mgcfd_solver *s, int level, double *rms);
/* check_for_invalid_variables(); synchronises; returns MGCFD_OK or MGCFD_ERR_NAN/NEG_*;
 * *bad_cell = first offending cell in original numbering          validation.cpp:107-138 */
int mgcfd_check_for_invalid_variables(mgcfd_solver *s, int level, int64_t *bad_cell);
/* What the checks INSIDE the launches issued so far have found (every time_step / fused stage carries the reference's
 * check_for_invalid_variables; the calls themselves are asynchronous): MGCFD_OK or the first failing launch's class and
 * cell.  Unlike mgcfd_check_for_invalid_variables it does not look at the current state — the reference checks after a
 * time_step only, so a value a prolongation spoils after the last sweep goes unnoticed there too.  Synchronises; clears the flag. */
int mgcfd_pending_invalid_state(mgcfd_solver *s, int64_t *bad_cell);
/* mg_restrict(variables[fine] -> variables[fine+1])                 mg_loops.cpp:30-202 */
int mgcfd_restrict(mgcfd_solver *s, int fine_level);
/* prolong_residuals_interpolate_proper(residuals[fine+1] -> variables[fine])   mg_loops.cpp:678-864 */
int mgcfd_prolong(mgcfd_solver *s, int fine_level);

/* ---------------------------------------------------------------------------------
 * Cycle driver — the state machine of src/euler3d_cpu_double.cpp:371-694
 * --------------------------------------------------------------------------------- */
/* `sweeps` smoothing sweeps on one level, no multigrid transfer: the per-level body of the cycle
 * loop (copy, step factor, RK x [fluxes, time_step], residual; euler3d_cpu_double.cpp:383-508). */
int mgcfd_smooth(mgcfd_solver *s, int level, int sweeps);
/* Runs `cycles` (multigrid) cycles from the solver's current state.  rms_out (may be NULL)
 * receives, per cycle, the level-0 RMS the reference prints.  Synchronises before returning.
 * On an invalid state returns MGCFD_ERR_NAN / NEG_* like the reference's exit().
 * The check runs inside every time_step launch; the earliest failing launch and, within it, the smallest original
 * cell id are what is reported (the reference stops at exactly that cell).  The cycles of the current batch (up to
 * 4096) still run to the end; rms_out entries from the failing cycle on are NaN. */
int mgcfd_run_cycles(mgcfd_solver *s, int cycles, double *rms_out);
/* Where the last MGCFD_ERR_NAN / NEG_DENSITY / NEG_ENERGY was found: *cell = original cell id (the reference's
 * "Cell %ld"), *cycle = 0-based cycle of the mgcfd_run_cycles call (-1: not known — graph replay, or found by another call). */
int mgcfd_invalid_state_location(const mgcfd_solver *s, int64_t *cell, int *cycle);

/* ---------------------------------------------------------------------------------
 * State access (synchronous; original numbering)
 * --------------------------------------------------------------------------------- */
int mgcfd_get_array(mgcfd_solver *s, int level, int which, double *out);        /* [nel*5] or [nel] */
int mgcfd_set_array(mgcfd_solver *s, int level, int which, const double *in);
/* Edge weights after adjust/dampen, original edge order: [n_edges] records. */
/* Device address of a node array as the library holds it: [ncols][stride] fp64, structure of arrays in the
 * LIBRARY's node numbering (ncols = 5, or 1 for step factors / volumes); *count = ncols * stride elements.
 * For moving a whole level's array between two solvers built from the SAME level data (same numbering), e.g. one
 * multigrid level per GPU: send the coarse `variables` after mgcfd_restrict, the coarse `residuals` before
 * mgcfd_prolong.  The address of `variables` / `old_variables` changes with every smoothing sweep (the state buffers
 * rotate): ask again after each one.  A caller that WRITES through the pointer must say so with
 * mgcfd_array_written before the next library call on that level. */
int mgcfd_array_devptr(mgcfd_solver *s, int level, int which, void **devptr, int64_t *count);
int mgcfd_array_written(mgcfd_solver *s, int level, int which);
int mgcfd_get_edges(mgcfd_solver *s, int level, mgcfd_edge *out);
/* One multigrid level per rank: `dev_src` is the next-coarser level's `variables` as ANOTHER solver built from the same
 * level data holds it after its mgcfd_restrict(fine_level) (its mgcfd_array_devptr, or a received copy of it).  Copies
 * the coarse nodes that HAVE children; a coarse node without children keeps this solver's value, as mg_restrict leaves
 * it (src/Kernels/mg_loops.cpp:63-78,174-189) — only the rank that sweeps the coarse level has that value. */
int mgcfd_accept_restricted(mgcfd_solver *s, int fine_level, const void *dev_src);

/* ---------------------------------------------------------------------------------
 * Monitoring — LoopNumIters.csv / Times.csv contents
 * (src/Monitoring/loop_stats.cpp:48-171, src/Monitoring/timer.cpp:58-195)
 * --------------------------------------------------------------------------------- */
int mgcfd_get_loop_iters(const mgcfd_solver *s, int level, int64_t out[MGCFD_NUM_LOOPS]);
int mgcfd_get_loop_times(mgcfd_solver *s, int level, double out_seconds[MGCFD_NUM_LOOPS]);
int mgcfd_reset_monitoring(mgcfd_solver *s);
/* Average GPU duration (seconds) of the internal-edge flux launches issued since the last
 * reset, measured with hipEvents on the launch stream, and how many launches that covers.
 * Requires MGCFD_OPT_TIMING. */
int mgcfd_get_flux_kernel_time(mgcfd_solver *s, int level, double *avg_seconds, int64_t *launches);

/* Diagnostic: mean GPU time of `launches` back-to-back flux launches (internal + boundary +
 * far field, starting from zero fluxes), hipEvents around the batch on the solver's stream. */
int mgcfd_bench_flux(mgcfd_solver *s, int level, int launches, double *avg_seconds);
/* The same for the indirect_rw probe (src/Kernels/indirect_rw_loop.cpp:8-78; fluxes += ..., accumulating over the
 * launches): the empirical data-movement ceiling of the flux kernel on this level's tiles. */
int mgcfd_bench_indirect_rw(mgcfd_solver *s, int level, int launches, double *avg_seconds);
/* ... and for a tile-shaped STREAM of exactly the bytes SURVEY.md §8(d) prices for that launch (40 B per internal edge + 40 B per
 * node read, 40 B per node written; one workgroup per tile, nothing dependent, nothing computed): the practical ceiling of its
 * data movement on this chip, launch included.  Overwrites `fluxes`. */
int mgcfd_bench_stream_ceiling(mgcfd_solver *s, int level, int launches, double *avg_seconds);

/* ---------------------------------------------------------------------------------
 * Multi-GPU hooks (one process per GPU; the collectives themselves are issued by the
 * host through RCCL — see INTEGRATION.md).  compute_step_factor's global min
 * (cfd_loops.cpp:137-150) is split so an all-reduce(min) can run between the halves.
 * --------------------------------------------------------------------------------- */
/* First half: per-node 0.5*dt and the rank-local minimum, left in a device scalar. */
int mgcfd_step_factor_local(mgcfd_solver *s, int level);
/* Device address of that fp64 scalar (for an in-place RCCL all-reduce MIN). */
int mgcfd_step_factor_min_devptr(mgcfd_solver *s, int level, void **devptr);
/* Second half: step_factors[i] = min_dt / volumes[i]. */
int mgcfd_step_factor_apply(mgcfd_solver *s, int level);
/* One smoothing sweep (as mgcfd_smooth) split around that all-reduce, with the fused stage
 * kernels: sweep_begin = first half of compute_step_factor (skipped when the launch that produced
 * the variables already left its minima behind) + reduction to the scalar behind
 * mgcfd_step_factor_min_devptr; [all-reduce MIN that scalar across ranks];
 * sweep_end = the RK stages (fluxes + time_step fused, "/ volume" applied in the first) + residual.
 * Optional, to HIDE the all-reduce: sweep_flux0 computes the first stage's fluxes, which do not
 * depend on the time step, and may run while the collective is in flight; sweep_end then starts
 * with time_step on them. */
int mgcfd_sweep_begin(mgcfd_solver *s, int level);
/* The same split with the PARTIAL minima as the exchanged quantity: *devptr = the level's per-workgroup minima
 * (*count fp64 values, a few KB), to be all-reduced (MIN, element-wise) in place of the scalar.  sweep_begin_partials
 * then launches nothing when an earlier launch already left the minima behind, and sweep_end_partials lets the first
 * stage take the minimum over the (now global) partials — one small kernel and 5 us less on the way to the collective.
 * Same results as sweep_begin / sweep_end. */
int mgcfd_step_factor_partials_devptr(mgcfd_solver *s, int level, void **devptr, int *count);
/* The Runge-Kutta stages of such a sweep ONE AT A TIME, for a partitioned level: after mgcfd_sweep_begin[_partials] and
 * the all-reduce, call mgcfd_sweep_stage(s, level, j, partials) for j = 0, 1, 2 — each a single fused launch (fluxes +
 * time_step; the first finishes compute_step_factor, the last writes the residual and ends the sweep) — and between
 * them move the ghosts' new values with mgcfd_halo_pack / _unpack on MGCFD_ARR_STAGE (the state that stage wrote; the
 * ghosts hold no rows, so a stage leaves them at the sweep's start state until the message arrives). */
int mgcfd_sweep_stage(mgcfd_solver *s, int level, int j, int partials);
int mgcfd_sweep_begin_partials(mgcfd_solver *s, int level);
int mgcfd_sweep_end_partials(mgcfd_solver *s, int level);
int mgcfd_sweep_flux0(mgcfd_solver *s, int level);
int mgcfd_sweep_end(mgcfd_solver *s, int level);
/* Halo exchange of a partitioned level.  A plan is a list of local node ids (the nodes this rank
 * sends to one peer, or the ghosts it receives from it, in an order both sides agree on);
 * pack copies their 5 values of array `which` (MGCFD_ARR_*) into a contiguous [n][5] fp64 message
 * in DEVICE memory, unpack writes a received message into them.  The message itself moves by
 * RCCL send/recv (torch.distributed) between the two calls. */
int mgcfd_halo_plan(mgcfd_solver *s, int level, int64_t n, const int64_t *node_ids, int *plan);
int mgcfd_halo_pack(mgcfd_solver *s, int level, int plan, int which, void *dev_buf);
int mgcfd_halo_unpack(mgcfd_solver *s, int level, int plan, int which, const void *dev_buf);
/* Sum of squared residuals of the level, left in a device scalar (all-reduce SUM, then
 * rms = sqrt(sum / global_nel)). */
int mgcfd_residual_sumsq(mgcfd_solver *s, int level, void **devptr);

/* ---------------------------------------------------------------------------------
 * Multi-GPU in the C++ host: a level PARTITIONED over ranks (mgcfd_create_partitioned), the whole
 * sweep loop inside the library.  The reference has no distributed path; what a partitioned level
 * needs follows from its loops: one all-reduce(MIN) of the time step per sweep
 * (src/Kernels/cfd_loops.cpp:137-150) and, because every RK stage reads the neighbours' new state
 * (src/Kernels/flux_loops.cpp:133-136 after cfd_loops.cpp:241-268), one halo message per neighbouring
 * rank after every stage.  Per stage the library runs the tiles next to ghost nodes first, packs every
 * peer's segment with ONE launch, sends (RCCL ncclSend/ncclRecv grouped on a second stream; between the
 * solvers of one process hipMemcpyPeerAsync over xGMI), runs the interior tiles while the message travels,
 * and unpacks with ONE launch before the next stage's boundary tiles.  Results equal the unpartitioned
 * level's bit for bit on owned nodes.  The solvers of one process skip the message buffers altogether: behind its
 * boundary tiles a rank stores the nodes its peers need straight into their ghost slots (one launch, peer access
 * over xGMI) and a peer's next stage waits for that launch's event (MGCFD_GROUP_DIRECT=0 in the environment keeps
 * the buffered form).
 *
 * Two ways to be a rank:
 *   one rank per PROCESS (RCCL):  mgcfd_rccl_unique_id on rank 0, the 128 bytes handed to every rank by the
 *       launcher (MPI, torch.distributed, a file), mgcfd_rank_attach_rccl, mgcfd_rank_set_halo,
 *       mgcfd_rank_exchange once, then mgcfd_rank_sweeps.  librccl is loaded when first needed.
 *   the solvers of ONE process (one per device): mgcfd_group_create, mgcfd_rank_set_halo on each,
 *       mgcfd_group_exchange once, then mgcfd_group_sweeps — what euler3d_gpu_double --gpus N runs.
 * --------------------------------------------------------------------------------- */
typedef struct mgcfd_group mgcfd_group;
int mgcfd_rccl_unique_id(void *out128);
int mgcfd_rank_attach_rccl(mgcfd_solver *s, int rank, int world, const void *id128);
int mgcfd_rank_detach(mgcfd_solver *s);
/* The level's neighbours: peers[k] ascending; send_ids[k] = the OWNED local nodes peer k holds as ghosts, recv_ids[k] =
 * the local GHOSTS peer k owns, both in an order the two ranks agree on (ascending global id).  Also splits the level's
 * tiles into boundary / interior for the overlapped exchange. */
int mgcfd_rank_set_halo(mgcfd_solver *s, int level, int n_peers, const int *peers, const int64_t *send_counts,
                        const int64_t *const *send_ids, const int64_t *recv_counts, const int64_t *const *recv_ids);
int mgcfd_rank_halo_info(const mgcfd_solver *s, int level, int64_t out[4]);   /* boundary tiles, interior tiles, nodes sent, nodes received */
/* What the solver is a rank of, as the library sees it: out[0] this rank, out[1] the number of ranks, out[2] the transport
 * (0 none, 1 an RCCL communicator, 2 an in-process group, 3 plain attachment: HIP IPC messages only), out[3] the size the RCCL
 * communicator itself reports (ncclCommCount; -1 without one). */
int mgcfd_rank_info(const mgcfd_solver *s, int out[4]);
/* MGCFD_OPT_GRAPH = 1 on an RCCL rank (mgcfd_rank_sweeps): out[0] = sweep graphs instantiated for the level (0..3, one per
 * buffer rotation), out[1] = 1 when a capture was refused — the sweeps then run call by call, with the same results —,
 * out[2] = sweeps replayed from a graph so far.  A caller that times the replayed form asks before it believes the figure. */
int mgcfd_rank_graph_status(const mgcfd_solver *s, int level, int64_t out[3]);
int mgcfd_rank_exchange(mgcfd_solver *s, int level);          /* ghosts of `variables` <- owners (after mgcfd_set_array) */
int mgcfd_rank_sweeps(mgcfd_solver *s, int level, int sweeps); /* the per-level body of the cycle loop, `sweeps` times; asynchronous */
int mgcfd_rank_residual_sumsq(mgcfd_solver *s, int level, double *sum_all_ranks);
/* Ranks in different processes, messages as direct stores (opt-in; rehearsed with two processes on ONE GPU only): every rank
 * publishes HIP IPC handles of its three state buffers and of a few flag words, opens its neighbours', and a stage's message
 * is then ONE launch that stores the nodes the neighbours need straight into their ghost slots (over xGMI between devices)
 * and raises their flags; the neighbour's next stage starts behind a one-wave launch that waits for them.  No message
 * buffers, no second stream, no RCCL call per stage; with EVERY rank attached the all-reduce of a global time step goes
 * through the same flags (every rank stores its minimum into every other rank's memory) and no collective library is needed
 * at all, otherwise it stays on RCCL.
 *   mgcfd_rank_attach_plain (or mgcfd_rank_attach_rccl), mgcfd_rank_set_halo, then mgcfd_rank_ipc_export_size / _export, the
 *   blobs handed round by the launcher, mgcfd_rank_ipc_attach (the other ranks' blobs in any order: at least the neighbours',
 *   at most 16 ranks); from then on mgcfd_rank_exchange / mgcfd_rank_sweeps run the direct form.
 *   mgcfd_rank_ipc_status: waits for a neighbour that gave up (about 2 s each) since the last call; 0 = all messages arrived.
 *   A wait that gave up is an ERROR, not a statistic: the stages behind it ran on stale ghosts.  While the count is not zero
 *   mgcfd_synchronize and mgcfd_get_array return MGCFD_ERR_HIP, and so does the next mgcfd_rank_sweeps after one of them has
 *   seen it; mgcfd_rank_ipc_status reads the count and thereby acknowledges it (mgcfd_rank_ipc_detach clears it too).
 *   The ranks keep each other in step through the flags only WITHIN this form: before the first mgcfd_rank_exchange — and
 *   whenever a rank has touched the level by other means (mgcfd_set_array, another kind of sweep) — the caller synchronises
 *   the ranks (a barrier), or a neighbour's stores may land in a buffer that is still in use. */
int mgcfd_rank_attach_plain(mgcfd_solver *s, int rank, int world);
int mgcfd_rank_ipc_export_size(mgcfd_solver *s, int level, int64_t *bytes);
int mgcfd_rank_ipc_export(mgcfd_solver *s, int level, void *out);
int mgcfd_rank_ipc_attach(mgcfd_solver *s, int level, int n_exports, const void *const *exports);
int mgcfd_rank_ipc_status(mgcfd_solver *s, int level, int *timed_out);
int mgcfd_rank_ipc_detach(mgcfd_solver *s, int level);       /* back to the buffered form; closes the neighbours' mappings */
int mgcfd_group_create(int n, mgcfd_solver *const *solvers, mgcfd_group **out);   /* solvers[r] becomes rank r of n */
void mgcfd_group_destroy(mgcfd_group *g);
int mgcfd_group_exchange(mgcfd_group *g, int level);
int mgcfd_group_sweeps(mgcfd_group *g, int level, int sweeps);   /* asynchronous; a host thread per rank issues that rank's launches */
/* The same with calc_rms (src/Kernels/validation.cpp:91-105) after every sweep — the reference's cycle loop prints it per
 * cycle, src/euler3d_cpu_double.cpp:383-508 — gathered on the devices and read back ONCE: rms_of_each[k] = RMS after
 * sweep k.  At most 4096 sweeps per call.  Synchronises. */
int mgcfd_group_sweeps_rms(mgcfd_group *g, int level, int sweeps, double *rms_of_each);
int mgcfd_group_rms(mgcfd_group *g, int level, double *rms);
/* V-cycles on a PARTITIONED HIERARCHY (mgcfd_create_partitioned_mg; mgcfd_rank_set_halo on EVERY level, ghosts current:
 * mgcfd_group_exchange / mgcfd_rank_exchange on every level once): the reference's cycle — sweeps on levels 0 .. n-1, n-2 .. 1,
 * mg_restrict on the way up, prolong_residuals_interpolate_proper on the way down, src/euler3d_cpu_double.cpp:371-694 — with
 * the ghost values moved where the next loop reads them: `variables` after every time_step (one message per Runge-Kutta stage,
 * as in mgcfd_group_sweeps), after mg_restrict (coarse ghosts) and after the prolongation (fine ghosts), the coarse `residuals`
 * before the prolongation; one all-reduce(MIN) of the time step per sweep.  Every level equals mgcfd_run_cycles on the whole
 * hierarchy bit for bit on owned nodes.  rms_out (may be NULL): the level-0 RMS after the level-0 sweep of each cycle, as the
 * reference prints it.  Returns MGCFD_OK or MGCFD_ERR_NAN / NEG_* (check_for_invalid_variables inside every time_step).
 * At most 4096 cycles per call.  Synchronises. */
int mgcfd_group_cycles(mgcfd_group *g, int cycles, double *rms_out);
int mgcfd_rank_cycles(mgcfd_solver *s, int cycles, double *rms_out);      /* one rank per process over RCCL (mgcfd_rank_attach_rccl) */
int mgcfd_group_synchronize(mgcfd_group *g);

#ifdef __cplusplus
}
#endif
#endif /* MGCFD_H */
