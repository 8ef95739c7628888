// ref_harness.cpp — OUR glue (test infrastructure) that exposes the REAL reference
// kernels through a C ABI so tests can pin the oracle against them.
//
// It is compiled by oracle/build_ref.sh together with the reference's own
// sources taken where they lie under /root/reference/src (never copied into
// this repository); the outputs go to oracle/_ref/ only.  It defines the
// globals the reference expects from its main() (src/Base/globals.h:4-16,
// src/euler3d_cpu_double.cpp:32-42) and forwards to the reference functions.
#include <omp.h>
#include <string.h>

#include "common.h"       // reference: src/Base/common.h
#include "io.h"           // reference: read_grid
#include "io_enhanced.h"  // reference: read_mg_connectivity, duplicate_mesh
#include "flux_loops.h"
#include "indirect_rw_loop.h"
#include "cfd_loops.h"
#include "mg_loops.h"
#include "validation.h"
#include "loop_stats.h"

int levels = 0;
int level = 0;
int current_kernel;
int mesh_variant;
double ff_variable[NVAR];
double3 ff_flux_contribution_momentum_x;
double3 ff_flux_contribution_momentum_y;
double3 ff_flux_contribution_momentum_z;
double3 ff_flux_contribution_density_energy;

namespace {
struct grid_t {
    long nel = 0, n_edges = 0, n_int = 0, n_bnd = 0, n_wall = 0, int_start = 0, bnd_start = 0, wall_start = 0;
    double *volumes = nullptr;
    edge_neighbour *edges = nullptr;
    double3 *coords = nullptr;
} g_grid;
}

extern "C" {

void ref_init(int n_levels, int variant)
{
    omp_set_num_threads(1);   // four loops carry unguarded "omp parallel for" 
    levels = n_levels;
    level = 0;
    mesh_variant = variant;
    set_config_defaults();
    init_iters();
    initialize_far_field_conditions();
}

void ref_set_level(int l) { level = l; }

// out[0..4]=ff_variable, then fc_mx, fc_my, fc_mz, fc_de (3 each)
void ref_get_farfield(double *out)
{
    for (int v = 0; v < NVAR; v++) out[v] = ff_variable[v];
    const double3 *src[4] = { &ff_flux_contribution_momentum_x, &ff_flux_contribution_momentum_y,
                              &ff_flux_contribution_momentum_z, &ff_flux_contribution_density_energy };
    for (int k = 0; k < 4; k++) { out[5 + 3 * k] = src[k]->x; out[6 + 3 * k] = src[k]->y; out[7 + 3 * k] = src[k]->z; }
}

void ref_compute_flux_edge(long first, long n, const edge_neighbour *edges, const double *variables, double *fluxes)
{ compute_flux_edge(first, n, edges, variables, fluxes); }

void ref_compute_boundary_flux_edge(long first, long n, const edge_neighbour *edges, const double *variables, double *fluxes)
{ compute_boundary_flux_edge(first, n, edges, variables, fluxes); }

void ref_compute_wall_flux_edge(long first, long n, const edge_neighbour *edges, const double *variables, double *fluxes)
{ compute_wall_flux_edge(first, n, edges, variables, fluxes); }

void ref_indirect_rw(long first, long n, const edge_neighbour *edges, const double *variables, double *fluxes)
{ indirect_rw(first, n, edges, variables, fluxes); }

void ref_compute_step_factor(long nel, const double *variables, const double *volumes, double *sf)
{ compute_step_factor(nel, variables, volumes, sf); }

void ref_compute_step_factor_legacy(long nel, const double *variables, const double *volumes, double *sf)
{ compute_step_factor_legacy(nel, variables, volumes, sf); }

void ref_time_step(int j, long nel, const double *sf, double *fluxes, const double *old_variables, double *variables)
{ time_step(j, nel, sf, fluxes, old_variables, variables); }

void ref_residual(long nel, const double *old_variables, const double *variables, double *residuals)
{ residual(nel, old_variables, variables, residuals); }

double ref_calc_rms(long nel, const double *residuals) { return calc_rms(nel, residuals); }

void ref_adjust_ewt(const double3 *coords, long n, edge_neighbour *edges) { adjust_ewt(coords, n, edges); }
void ref_dampen_ewt(long n, edge_neighbour *edges, double f) { dampen_ewt(n, edges, f); }

void ref_mg_restrict(double *v1, double *v2, long nel2, long *mapping, long *up_scratch, long mgc)
{ mg_restrict(v1, v2, nel2, mapping, up_scratch, mgc); }

void ref_prolong_residuals_interpolate_proper(edge_neighbour *edges, long num_edges, double *residuals1,
        double *residuals2, double *variables2, long nel2, long *mapping, double3 *coords1, double3 *coords2)
{ prolong_residuals_interpolate_proper(edges, num_edges, residuals1, residuals2, variables2, nel2, mapping, coords1, coords2); }

// read_grid: first call parses and returns the sizes (out[8]); ref_grid_copy hands the arrays over.
void ref_read_grid(const char *path, long *out)
{
    read_grid(path, &g_grid.nel, &g_grid.volumes, &g_grid.n_edges, &g_grid.n_int, &g_grid.n_bnd, &g_grid.n_wall,
              &g_grid.int_start, &g_grid.bnd_start, &g_grid.wall_start, &g_grid.edges, &g_grid.coords);
    out[0] = g_grid.nel; out[1] = g_grid.n_edges; out[2] = g_grid.n_int; out[3] = g_grid.n_bnd;
    out[4] = g_grid.n_wall; out[5] = g_grid.int_start; out[6] = g_grid.bnd_start; out[7] = g_grid.wall_start;
}

void ref_grid_copy(double *volumes, edge_neighbour *edges, double3 *coords)
{
    memcpy(volumes, g_grid.volumes, sizeof(double) * g_grid.nel);
    memcpy(edges, g_grid.edges, sizeof(edge_neighbour) * g_grid.n_edges);
    memcpy(coords, g_grid.coords, sizeof(double3) * g_grid.nel);
    dealloc<double>(g_grid.volumes);
    dealloc<edge_neighbour>(g_grid.edges);
    dealloc<double3>(g_grid.coords);
    g_grid = grid_t();
}

} // extern "C"
