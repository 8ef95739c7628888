/*
 * mgcfd_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's edge-flux / multigrid hot path with
 * the same floating-point operation order, so that built with
 * `-fno-fast-math -ffp-contract=off` it reproduces the reference compiled the
 * same way bit for bit (checked by tests/test_oracle_vs_reference.py when
 * oracle/_ref/ is present, and against tests/golden/ everywhere).
 * See mgcfd_oracle.h for the rules on who may use this file.
 */
#include "mgcfd_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define GAMMA 1.4                 /* src/Base/const.h:9 */
#define FF_MACH 1.2               /* src/Base/const.h:14 */
#define DEG_ANGLE_OF_ATTACK 0.0   /* src/Base/const.h:15 */

/* src/Base/common.h:24: a float literal widened to double (0.20000000298...). */
static const double k_smoothing = (double)0.2f;

/* ------------------------------------------------------------------------ */
/* Point state derived from the 5 conserved variables.                       */
/* cfd_loops.h:121-148 (velocity, |v|^2, pressure, speed of sound) and        */
/* cfd_loops.h:57-83 (the four flux-contribution vectors).                    */
/* ------------------------------------------------------------------------ */
typedef struct {
    double rho, mx, my, mz, en;
    double vx, vy, vz;
    double speed_sqd, pressure;
    ora_vec3 fmx, fmy, fmz, fde;
} point_state;

static inline void load_point(const double *variables, int64_t node, point_state *s)
{
    const double *q = variables + node * ORA_NVAR;
    s->rho = q[0]; s->mx = q[1]; s->my = q[2]; s->mz = q[3]; s->en = q[4];
    s->vx = s->mx / s->rho;                     /* compute_velocity, cfd_loops.h:121-126 */
    s->vy = s->my / s->rho;
    s->vz = s->mz / s->rho;
    s->speed_sqd = s->vx * s->vx + s->vy * s->vy + s->vz * s->vz; /* :135-138 */
    /* compute_pressure, :140-143: (GAMMA-1)*(E - 0.5*rho*|v|^2), left-assoc. */
    s->pressure = (GAMMA - 1.0) * (s->en - 0.5 * s->rho * s->speed_sqd);
}

static inline double sound_speed(const point_state *s)
{
    return sqrt(GAMMA * s->pressure / s->rho);  /* cfd_loops.h:145-148 */
}

static inline void flux_contributions(double mx, double my, double mz, double en,
                                      double pressure, double vx, double vy, double vz,
                                      ora_vec3 *fmx, ora_vec3 *fmy, ora_vec3 *fmz, ora_vec3 *fde)
{
    /* cfd_loops.h:57-83 */
    fmx->x = vx * mx + pressure;
    fmx->y = vx * my;
    fmx->z = vx * mz;
    fmy->x = fmx->y;
    fmy->y = vy * my + pressure;
    fmy->z = vy * mz;
    fmz->x = fmx->z;
    fmz->y = fmy->z;
    fmz->z = vz * mz + pressure;
    double de_p = en + pressure;
    fde->x = vx * de_p;
    fde->y = vy * de_p;
    fde->z = vz * de_p;
}

static inline void point_fluxes(point_state *s)
{
    flux_contributions(s->mx, s->my, s->mz, s->en, s->pressure, s->vx, s->vy, s->vz,
                       &s->fmx, &s->fmy, &s->fmz, &s->fde);
}

void ora_far_field(ora_farfield *ff)
{
    /* cfd_loops.h:85-119 */
    const double angle = (double)(3.1415926535897931 / 180.0) * (double)(DEG_ANGLE_OF_ATTACK);
    ff->var[0] = 1.4;
    double ff_pressure = 1.0;
    double ff_c = sqrt(GAMMA * ff_pressure / ff->var[0]);
    double ff_speed = (double)(FF_MACH) * ff_c;
    double vx = ff_speed * cos(angle);
    double vy = ff_speed * sin(angle);
    double vz = 0.0;
    ff->var[1] = ff->var[0] * vx;
    ff->var[2] = ff->var[0] * vy;
    ff->var[3] = ff->var[0] * vz;
    ff->var[4] = ff->var[0] * (0.5 * (ff_speed * ff_speed)) + (ff_pressure / (GAMMA - 1.0));
    flux_contributions(ff->var[1], ff->var[2], ff->var[3], ff->var[4], ff_pressure, vx, vy, vz,
                       &ff->fc_mx, &ff->fc_my, &ff->fc_mz, &ff->fc_de);
}

void ora_initialize_variables(int64_t nel, double *variables, const ora_farfield *ff)
{
    for (int64_t i = 0; i < nel; i++)
        for (int v = 0; v < ORA_NVAR; v++)
            variables[i * ORA_NVAR + v] = ff->var[v];
}

/* ------------------------------------------------------------------------ */
/* Edge fluxes                                                               */
/* ------------------------------------------------------------------------ */
void ora_compute_flux_edge(int64_t first_edge, int64_t nedges, const ora_edge *edges,
                           const double *variables, double *fluxes)
{
    /* Loop: flux_loops.cpp:133-136; body: flux_kernel.elemfunc.c:18-229
     * (default build: no FLUX_REUSE_*, no FLUX_PRECOMPUTE_EDGE_WEIGHTS). */
    for (int64_t e = first_edge; e < first_edge + nedges; e++) {
        const int64_t a = edges[e].a, b = edges[e].b;
        const double ex = edges[e].x, ey = edges[e].y, ez = edges[e].z;
        const double ewt = sqrt(ex * ex + ey * ey + ez * ez);          /* :27 */

        point_state B, A;
        load_point(variables, b, &B);                                    /* :35-77 */
        double speed_b = sqrt(B.speed_sqd);
        double c_b = sound_speed(&B);
        point_fluxes(&B);
        load_point(variables, a, &A);                                    /* :84-128 */
        double speed_a = sqrt(A.speed_sqd);
        double c_a = sound_speed(&A);
        point_fluxes(&A);

        /* :130-136 (factor_b is the same expression) */
        double factor_a = -ewt * k_smoothing * 0.5 * (speed_a + speed_b + c_a + c_b);
        double factor_b = -ewt * k_smoothing * 0.5 * (speed_a + speed_b + c_a + c_b);
        double fx = -0.5 * ex, fy = -0.5 * ey, fz = -0.5 * ez;           /* :138-140 */

        /* :142-161 */
        double p_a  = factor_a * (A.rho - B.rho) + fx * (A.mx + B.mx) + fy * (A.my + B.my) + fz * (A.mz + B.mz);
        double pe_a = factor_a * (A.en - B.en) + fx * (A.fde.x + B.fde.x) + fy * (A.fde.y + B.fde.y) + fz * (A.fde.z + B.fde.z);
        double mx_a = factor_a * (A.mx - B.mx) + fx * (A.fmx.x + B.fmx.x) + fy * (A.fmx.y + B.fmx.y) + fz * (A.fmx.z + B.fmx.z);
        double my_a = factor_a * (A.my - B.my) + fx * (A.fmy.x + B.fmy.x) + fy * (A.fmy.y + B.fmy.y) + fz * (A.fmy.z + B.fmy.z);
        double mz_a = factor_a * (A.mz - B.mz) + fx * (A.fmz.x + B.fmz.x) + fy * (A.fmz.y + B.fmz.y) + fz * (A.fmz.z + B.fmz.z);
        /* :170-189 */
        double p_b  = factor_b * (B.rho - A.rho) - fx * (A.mx + B.mx) - fy * (A.my + B.my) - fz * (A.mz + B.mz);
        double pe_b = factor_b * (B.en - A.en) - fx * (A.fde.x + B.fde.x) - fy * (A.fde.y + B.fde.y) - fz * (A.fde.z + B.fde.z);
        double mx_b = factor_b * (B.mx - A.mx) - fx * (A.fmx.x + B.fmx.x) - fy * (A.fmx.y + B.fmx.y) - fz * (A.fmx.z + B.fmx.z);
        double my_b = factor_b * (B.my - A.my) - fx * (A.fmy.x + B.fmy.x) - fy * (A.fmy.y + B.fmy.y) - fz * (A.fmy.z + B.fmy.z);
        double mz_b = factor_b * (B.mz - A.mz) - fx * (A.fmz.x + B.fmz.x) - fy * (A.fmz.y + B.fmz.y) - fz * (A.fmz.z + B.fmz.z);

        /* :218-228 */
        double *fa = fluxes + a * ORA_NVAR, *fb = fluxes + b * ORA_NVAR;
        fa[0] += p_a; fa[1] += mx_a; fa[2] += my_a; fa[3] += mz_a; fa[4] += pe_a;
        fb[0] += p_b; fb[1] += mx_b; fb[2] += my_b; fb[3] += mz_b; fb[4] += pe_b;
    }
}

void ora_compute_boundary_flux_edge(int64_t first_edge, int64_t nedges, const ora_edge *edges,
                                    const double *variables, double *fluxes)
{
    /* flux_loops.cpp:33-36 + flux_boundary_kernel.elemfunc.c:15-65.
     * Neighbour code -1: solid wall, pressure force only. */
    for (int64_t e = first_edge; e < first_edge + nedges; e++) {
        const int64_t b = edges[e].b;
        point_state B;
        load_point(variables, b, &B);
        double *fb = fluxes + b * ORA_NVAR;
        fb[0] += 0.0;
        fb[1] += edges[e].x * B.pressure;
        fb[2] += edges[e].y * B.pressure;
        fb[3] += edges[e].z * B.pressure;
        fb[4] += 0.0;
    }
}

void ora_compute_wall_flux_edge(int64_t first_edge, int64_t nedges, const ora_edge *edges,
                                const double *variables, double *fluxes, const ora_farfield *ff)
{
    /* flux_loops.cpp:67-70 + flux_wall_kernel.elemfunc.c:15-89.
     * Neighbour code -2: far-field flux against the ff_* constants. */
    for (int64_t e = first_edge; e < first_edge + nedges; e++) {
        const int64_t b = edges[e].b;
        point_state B;
        load_point(variables, b, &B);
        point_fluxes(&B);
        double fx = 0.5 * edges[e].x, fy = 0.5 * edges[e].y, fz = 0.5 * edges[e].z;
        double p  = fx * (ff->var[1] + B.mx) + fy * (ff->var[2] + B.my) + fz * (ff->var[3] + B.mz);
        double pe = fx * (ff->fc_de.x + B.fde.x) + fy * (ff->fc_de.y + B.fde.y) + fz * (ff->fc_de.z + B.fde.z);
        double mx = fx * (ff->fc_mx.x + B.fmx.x) + fy * (ff->fc_mx.y + B.fmx.y) + fz * (ff->fc_mx.z + B.fmx.z);
        double my = fx * (ff->fc_my.x + B.fmy.x) + fy * (ff->fc_my.y + B.fmy.y) + fz * (ff->fc_my.z + B.fmy.z);
        double mz = fx * (ff->fc_mz.x + B.fmz.x) + fy * (ff->fc_mz.y + B.fmz.y) + fz * (ff->fc_mz.z + B.fmz.z);
        double *fb = fluxes + b * ORA_NVAR;
        fb[0] += p; fb[1] += mx; fb[2] += my; fb[3] += mz; fb[4] += pe;
    }
}

void ora_indirect_rw(int64_t first_edge, int64_t nedges, const ora_edge *edges,
                     const double *variables, double *fluxes)
{
    /* indirect_rw_loop.cpp:62-65 + indirect_rw_kernel.elemfunc.c:4-94 */
    for (int64_t e = first_edge; e < first_edge + nedges; e++) {
        const int64_t a = edges[e].a, b = edges[e].b;
        const double *qa = variables + a * ORA_NVAR, *qb = variables + b * ORA_NVAR;
        double *fa = fluxes + a * ORA_NVAR, *fb = fluxes + b * ORA_NVAR;
        fa[0] += qb[0] + edges[e].x;
        fa[1] += qb[1] + edges[e].z;
        fa[2] += qb[2];
        fa[3] += qb[3];
        fa[4] += qb[4] + edges[e].y;
        fb[0] += qa[0]; fb[1] += qa[1]; fb[2] += qa[2]; fb[3] += qa[3]; fb[4] += qa[4];
    }
}

/* ------------------------------------------------------------------------ */
/* Node sweeps                                                               */
/* ------------------------------------------------------------------------ */
void ora_compute_step_factor(int64_t nel, const double *variables, const double *volumes,
                             double *step_factors)
{
    /* cfd_loops.cpp:98-125 */
    for (int64_t i = 0; i < nel; i++) {
        point_state s;
        load_point(variables, i, &s);
        double c = sound_speed(&s);
        double dt = cbrt(volumes[i]) / (sqrt(s.speed_sqd) + c);
        step_factors[i] = 0.5 * dt;
    }
    /* "Sync dt", cfd_loops.cpp:137-156: one global time step, then / volume. */
    double min_dt = step_factors[0];
    for (int64_t i = 0; i < nel; i++)
        if (step_factors[i] < min_dt) min_dt = step_factors[i];
    for (int64_t i = 0; i < nel; i++) step_factors[i] = min_dt;
    for (int64_t i = 0; i < nel; i++) step_factors[i] /= volumes[i];
}

void ora_compute_step_factor_legacy(int64_t nel, const double *variables, const double *volumes,
                                    double *step_factors)
{
    /* cfd_loops.cpp:37-61 (Rodinia's local step, used for mesh_name=fvcorr) */
    for (int64_t i = 0; i < nel; i++) {
        point_state s;
        load_point(variables, i, &s);
        double c = sound_speed(&s);
        step_factors[i] = 0.5 / (sqrt(volumes[i]) * (sqrt(s.speed_sqd) + c));
    }
}

void ora_time_step(int j, int64_t nel, const double *step_factors, double *fluxes,
                   const double *old_variables, double *variables)
{
    /* cfd_loops.cpp:241-268 */
    for (int64_t i = 0; i < nel; i++) {
        double factor = step_factors[i] / (double)(ORA_RK + 1 - j);
        for (int v = 0; v < ORA_NVAR; v++) {
            int64_t k = i * ORA_NVAR + v;
            variables[k] = old_variables[k] + factor * fluxes[k];
        }
        for (int v = 0; v < ORA_NVAR; v++) fluxes[i * ORA_NVAR + v] = 0.0;
    }
}

void ora_zero_fluxes(int64_t nel, double *fluxes)
{
    for (int64_t k = 0; k < nel * ORA_NVAR; k++) fluxes[k] = 0.0;
}

void ora_residual(int64_t nel, const double *old_variables, const double *variables,
                  double *residuals)
{
    for (int64_t k = 0; k < nel * ORA_NVAR; k++) residuals[k] = variables[k] - old_variables[k];
}

double ora_calc_rms(int64_t nel, const double *residuals)
{
    double rms = 0.0;
    for (int64_t k = 0; k < nel * ORA_NVAR; k++) rms += pow(residuals[k], 2);
    rms /= (double)nel;
    return sqrt(rms);
}

void ora_adjust_ewt(const ora_vec3 *coords, int64_t num_edges, ora_edge *edges)
{
    for (int64_t i = 0; i < num_edges; i++) {
        int64_t a = edges[i].a, b = edges[i].b;
        if (a >= 0 && b >= 0) {
            double dist = 0.0, d;
            d = coords[b].x - coords[a].x; dist += d * d;
            d = coords[b].y - coords[a].y; dist += d * d;
            d = coords[b].z - coords[a].z; dist += d * d;
            dist = sqrt(dist);
            edges[i].x /= dist; edges[i].y /= dist; edges[i].z /= dist;
        }
    }
}

void ora_dampen_ewt(int64_t num_edges, ora_edge *edges, double damping_factor)
{
    for (int64_t i = 0; i < num_edges; i++) {
        edges[i].x *= damping_factor; edges[i].y *= damping_factor; edges[i].z *= damping_factor;
    }
}

int ora_check_for_invalid_variables(const double *variables, int64_t n, int64_t *bad_cell)
{
    for (int64_t i = 0; i < n; i++) {
        for (int v = 0; v < ORA_NVAR; v++) {
            double x = variables[i * ORA_NVAR + v];
            if (isnan(x) || isinf(x)) { if (bad_cell) *bad_cell = i; return 1; }
        }
        if (variables[i * ORA_NVAR + 0] < 0.0) { if (bad_cell) *bad_cell = i; return 2; }
        if (variables[i * ORA_NVAR + 4] < 0.0) { if (bad_cell) *bad_cell = i; return 3; }
    }
    return 0;
}

int64_t ora_identify_differences(const double *test_values, const double *master_values,
                                 int64_t n, int mesh_variant)
{
    /* validation.cpp:159-197: rel 1e-8 with an absolute floor of 3e-19 (1e-15 for fvcorr). */
    const double rel = 10.0e-9;
    double abs_thresh = (mesh_variant == ORA_MESH_FVCORR) ? 1.0e-15 : 3.0e-19;
    for (int64_t k = 0; k < n * ORA_NVAR; k++) {
        double ok = master_values[k] * rel;
        if (ok < 0.0) ok *= -1.0;
        if (ok < abs_thresh) ok = abs_thresh;
        double diff = test_values[k] - master_values[k];
        if (diff < 0.0) diff *= -1.0;
        if (diff > ok) return k;
    }
    return -1;
}

/* ------------------------------------------------------------------------ */
/* Multigrid transfer                                                        */
/* ------------------------------------------------------------------------ */
void ora_mg_restrict(const double *variables1, double *variables2, int64_t nel2,
                     const int64_t *mapping, int64_t *up_scratch, int64_t mgc)
{
    /* mg_loops.cpp:63-78: zero only coarse nodes that have a child */
    for (int64_t i = 0; i < mgc; i++)
        for (int v = 0; v < ORA_NVAR; v++) variables2[mapping[i] * ORA_NVAR + v] = 0.0;
    for (int64_t i = 0; i < nel2; i++) up_scratch[i] = 0;          /* :94-98 */
    for (int64_t i = 0; i < mgc; i++) {                              /* :119-142 */
        int64_t p2 = mapping[i];
        for (int v = 0; v < ORA_NVAR; v++)
            variables2[p2 * ORA_NVAR + v] += variables1[i * ORA_NVAR + v];
        up_scratch[p2]++;
    }
    for (int64_t i = 0; i < nel2; i++) {                             /* :174-189 */
        double average = up_scratch[i] == 0 ? 1.0 : 1.0 / (double)up_scratch[i];
        for (int v = 0; v < ORA_NVAR; v++) variables2[i * ORA_NVAR + v] *= average;
    }
}

static inline double inv_dist(ora_vec3 p, ora_vec3 q)
{
    double dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
    return 1.0 / sqrt(dx * dx + dy * dy + dz * dz);
}

void ora_prolong_residuals_interpolate_proper(const ora_edge *edges, int64_t num_edges,
        const double *residuals1, const double *residuals2, double *variables2, int64_t nel2,
        const int64_t *mapping, const ora_vec3 *coords1, const ora_vec3 *coords2)
{
    /* mg_loops.cpp:701-709 */
    double *w_sums = (double *)calloc((size_t)nel2, sizeof(double));
    double *wavg = (double *)calloc((size_t)nel2 * ORA_NVAR, sizeof(double));

    /* mg_loops.cpp:730-812.  Level "1" = coarse (above), level "2" = fine. */
    for (int64_t e = 0; e < num_edges; e++) {
        const int64_t a2 = edges[e].a, b2 = edges[e].b;
        const int64_t a1 = mapping[a2], b1 = mapping[b2];
        const ora_vec3 ca1 = coords1[a1], ca2 = coords2[a2];
        const ora_vec3 cb1 = coords1[b1], cb2 = coords2[b2];
        const double *ra1 = residuals1 + a1 * ORA_NVAR, *rb1 = residuals1 + b1 * ORA_NVAR;
        double *wa = wavg + a2 * ORA_NVAR, *wb = wavg + b2 * ORA_NVAR;

        if ((ca2.x - ca1.x) == 0.0 && (ca2.y - ca1.y) == 0.0 && (ca2.z - ca1.z) == 0.0) {
            for (int v = 0; v < ORA_NVAR; v++) wa[v] = ra1[v];      /* :745-752 */
            w_sums[a2] = 1.0;
        } else {
            double w = inv_dist(ca2, ca1);                           /* :754-761 */
            for (int v = 0; v < ORA_NVAR; v++) wa[v] += w * ra1[v];
            w_sums[a2] += w;
            w = inv_dist(cb1, ca2);                                  /* :763-775 */
            for (int v = 0; v < ORA_NVAR; v++) wa[v] += w * rb1[v];
            w_sums[a2] += w;
        }

        if ((cb2.x - cb1.x) == 0.0 && (cb2.y - cb1.y) == 0.0 && (cb2.z - cb1.z) == 0.0) {
            for (int v = 0; v < ORA_NVAR; v++) wb[v] = rb1[v];      /* :783-790 */
            w_sums[b2] = 1.0;
        } else {
            double w = inv_dist(cb2, cb1);                           /* :792-799 */
            for (int v = 0; v < ORA_NVAR; v++) wb[v] += w * rb1[v];
            w_sums[b2] += w;
            /* :801-810 — the reference weights by dist(a1,b2) but reads
             * residuals1 of b1 (not a1).  Reproduced on purpose. */
            w = inv_dist(ca1, cb2);
            for (int v = 0; v < ORA_NVAR; v++) wb[v] += w * rb1[v];
            w_sums[b2] += w;
        }
    }

    /* mg_loops.cpp:844-852 */
    for (int64_t i = 0; i < nel2; i++) {
        for (int v = 0; v < ORA_NVAR; v++) {
            int64_t k = i * ORA_NVAR + v;
            wavg[k] /= w_sums[i];
            variables2[k] += residuals2[k] - wavg[k];
        }
    }
    free(w_sums);
    free(wavg);
}

/* ------------------------------------------------------------------------ */
/* File boundary                                                             */
/* ------------------------------------------------------------------------ */
int ora_read_grid(const char *path, int mesh_variant, int read_coords, ora_level *out)
{
    /* src/Base/io.cpp:56-177 */
    memset(out, 0, sizeof(*out));
    FILE *f = fopen(path, "r");
    if (!f) return 1;
    FILE *fc = NULL;
    if (read_coords) {
        size_t n = strlen(path) + 8;
        char *cpath = (char *)malloc(n);
        snprintf(cpath, n, "%s.coords", path);
        fc = fopen(cpath, "r");
        free(cpath);
        if (!fc) { fclose(f); return 2; }
    }
    long nel = 0, n_edges = 0;
    if (fscanf(f, "%ld %ld", &nel, &n_edges) != 2) { fclose(f); if (fc) fclose(fc); return 3; }
    out->nel = nel;
    out->n_edges = n_edges;
    out->volumes = (double *)malloc(sizeof(double) * (size_t)nel);
    out->coords = (ora_vec3 *)calloc((size_t)nel, sizeof(ora_vec3));
    ora_edge *bin = (ora_edge *)malloc(sizeof(ora_edge) * (size_t)(n_edges > 0 ? n_edges : 1));
    char *kind = (char *)malloc((size_t)(n_edges > 0 ? n_edges : 1));
    int64_t count = 0, n_int = 0, n_bnd = 0, n_wall = 0;
    int rc = 0;
    for (long i = 0; i < nel && !rc; i++) {
        int degree = 0;
        if (fscanf(f, "%lf %d", &out->volumes[i], &degree) != 2) { rc = 4; break; }
        if (fc && fscanf(fc, "%lf %lf %lf", &out->coords[i].x, &out->coords[i].y, &out->coords[i].z) != 3) { rc = 5; break; }
        for (int j = 0; j < degree; j++) {
            long nb; double wx, wy, wz;
            if (fscanf(f, "%ld %lf %lf %lf", &nb, &wx, &wy, &wz) != 4) { rc = 6; break; }
            if (nb < i) {                     /* io.cpp:93: each edge recorded from its higher end */
                if (count >= n_edges) { rc = 7; break; }
                if (nb == -1) { kind[count] = 1; n_bnd++; }
                else if (nb == -2) { kind[count] = 2; n_wall++; }
                else { kind[count] = 0; n_int++; }
                /* io.cpp:117-133: fvcorr flips every normal, others only internal ones */
                if (mesh_variant == ORA_MESH_FVCORR || nb >= 0) { wx *= -1; wy *= -1; wz *= -1; }
                bin[count].a = nb; bin[count].b = i;
                bin[count].x = wx; bin[count].y = wy; bin[count].z = wz;
                count++;
            }
        }
    }
    fclose(f);
    if (fc) fclose(fc);
    if (rc) { free(bin); free(kind); return rc; }

    /* io.cpp:149-177: [internal | boundary | wall], file order inside each class */
    out->n_internal = n_int; out->n_boundary = n_bnd; out->n_wall = n_wall;
    out->internal_start = 0; out->boundary_start = n_int; out->wall_start = n_int + n_bnd;
    out->edges = (ora_edge *)malloc(sizeof(ora_edge) * (size_t)(n_edges > 0 ? n_edges : 1));
    int64_t pos[3] = { 0, n_int, n_int + n_bnd };
    for (int64_t e = 0; e < count; e++) out->edges[pos[(int)kind[e]]++] = bin[e];
    for (int64_t e = count; e < n_edges; e++) { out->edges[e].a = -5; out->edges[e].b = -5; }
    free(bin); free(kind);
    return 0;
}

static int edge_before(const void *pp, const void *qq)
{
    /* compare_two_edges, src/Base/common.h:145-157, as a three-way comparison for qsort */
    const ora_edge *p = (const ora_edge *)pp, *q = (const ora_edge *)qq;
    if (p->a != q->a) return p->a < q->a ? -1 : 1;
    if (p->b != q->b) return p->b < q->b ? -1 : 1;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    if (p->y != q->y) return p->y < q->y ? -1 : 1;
    if (p->z != q->z) return p->z < q->z ? -1 : 1;
    return 0;
}

void ora_sort_edges_legacy(ora_level *L)
{
    qsort(L->edges + L->internal_start, (size_t)L->n_internal, sizeof(ora_edge), edge_before);
    qsort(L->edges + L->boundary_start, (size_t)L->n_boundary, sizeof(ora_edge), edge_before);
    qsort(L->edges + L->wall_start, (size_t)L->n_wall, sizeof(ora_edge), edge_before);
}

int ora_read_mg_connectivity(const char *path, int64_t **map, int64_t *mgc)
{
    FILE *f = fopen(path, "r");
    if (!f) return 1;
    long n = 0;
    if (fscanf(f, "%ld", &n) != 1) { fclose(f); return 2; }
    int64_t *m = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    for (long i = 0; i < n; i++) {
        long v;
        if (fscanf(f, "%ld", &v) != 1) { free(m); fclose(f); return 3; }
        m[i] = v;
    }
    fclose(f);
    *map = m; *mgc = n;
    return 0;
}

void ora_duplicate_mesh(ora_level *L, int m, int64_t nel_above)
{
    /* src/Base/io_enhanced.cpp:89-201: m disjoint copies; each edge class stays
     * contiguous ([m x internal][m x boundary][m x wall]); maps shift by nel_above. */
    if (m <= 1) return;
    const int64_t nel = L->nel;
    double *vol = (double *)malloc(sizeof(double) * (size_t)(nel * m));
    ora_vec3 *crd = (ora_vec3 *)malloc(sizeof(ora_vec3) * (size_t)(nel * m));
    for (int c = 0; c < m; c++)
        for (int64_t p = 0; p < nel; p++) { vol[c * nel + p] = L->volumes[p]; crd[c * nel + p] = L->coords[p]; }
    const int64_t counts[3] = { L->n_internal, L->n_boundary, L->n_wall };
    const int64_t starts[3] = { L->internal_start, L->boundary_start, L->wall_start };
    ora_edge *ed = (ora_edge *)malloc(sizeof(ora_edge) * (size_t)(L->n_edges * m));
    int64_t j = 0;
    for (int k = 0; k < 3; k++) {
        int64_t target = starts[k] * m;
        for (; j < target; j++) { ed[j].a = -5; ed[j].b = -5; }
        for (int c = 0; c < m; c++)
            for (int64_t e = 0; e < counts[k]; e++, j++) {
                ed[j] = L->edges[starts[k] + e];
                if (ed[j].a >= 0) ed[j].a += nel * c;
                if (ed[j].b >= 0) ed[j].b += nel * c;
            }
    }
    for (; j < L->n_edges * m; j++) { ed[j].a = -5; ed[j].b = -5; }
    if (L->mg_map) {
        int64_t *map = (int64_t *)malloc(sizeof(int64_t) * (size_t)(L->mgc * m));
        for (int c = 0; c < m; c++)
            for (int64_t n = 0; n < L->mgc; n++) map[c * L->mgc + n] = L->mg_map[n] + nel_above * c;
        free(L->mg_map);
        L->mg_map = map;
        L->mgc *= m;
    }
    free(L->volumes); L->volumes = vol;
    free(L->coords); L->coords = crd;
    free(L->edges); L->edges = ed;
    L->nel *= m; L->n_edges *= m;
    L->n_internal *= m; L->n_boundary *= m; L->n_wall *= m;
    L->boundary_start *= m; L->wall_start *= m;
}

void ora_alloc_state(ora_level *L)
{
    size_t n5 = (size_t)L->nel * ORA_NVAR;
    free(L->variables); free(L->old_variables); free(L->residuals); free(L->fluxes); free(L->step_factors);
    L->variables = (double *)calloc(n5, sizeof(double));
    L->old_variables = (double *)calloc(n5, sizeof(double));
    L->residuals = (double *)calloc(n5, sizeof(double));
    L->fluxes = (double *)calloc(n5, sizeof(double));
    L->step_factors = (double *)calloc((size_t)L->nel, sizeof(double));
}

void ora_free_level(ora_level *L)
{
    free(L->volumes); free(L->coords); free(L->edges); free(L->mg_map);
    free(L->variables); free(L->old_variables); free(L->residuals); free(L->fluxes); free(L->step_factors);
    memset(L, 0, sizeof(*L));
}

/* ------------------------------------------------------------------------ */
/* Driver: src/euler3d_cpu_double.cpp:321-694                                */
/* ------------------------------------------------------------------------ */
int ora_solve(ora_level *levels, int nlevels, int mesh_variant, int cycles,
              int run_indirect_rw, double *rms_out, ora_iters *iters)
{
    ora_farfield ff;
    ora_far_field(&ff);                                             /* :321 */
    /* :323 sizes this by nel[0]; mg_restrict indexes it by COARSE node, so a hierarchy whose coarse level is the
     * larger one overruns it there.  Sized by the largest level here: same results wherever the reference is defined. */
    int64_t scratch_n = levels[0].nel;
    for (int l = 1; l < nlevels; l++) if (levels[l].nel > scratch_n) scratch_n = levels[l].nel;
    int64_t *up_scratch = (int64_t *)malloc(sizeof(int64_t) * (size_t)scratch_n);
    for (int l = 0; l < nlevels; l++) {
        ora_initialize_variables(levels[l].nel, levels[l].variables, &ff);
        ora_zero_fluxes(levels[l].nel, levels[l].fluxes);
        ora_zero_fluxes(levels[l].nel, levels[l].residuals);
    }
    double damping = 0.0;                                           /* :337-352 */
    if (mesh_variant == ORA_MESH_M6_WING) damping = 5e-8;
    else if (mesh_variant == ORA_MESH_LA_CASCADE) damping = 1e-7;
    else if (mesh_variant == ORA_MESH_ROTOR_37) damping = 2e-7;
    if (damping != 0.0)
        for (int l = 0; l < nlevels; l++) {
            ora_adjust_ewt(levels[l].coords, levels[l].n_edges, levels[l].edges);
            ora_dampen_ewt(levels[l].n_edges, levels[l].edges, damping);
        }
    if (iters) memset(iters, 0, sizeof(ora_iters) * (size_t)nlevels);

    int level = 0, going_up = 1, rc = 0;
    for (int cyc = 0; cyc < cycles && !rc;) {                       /* :371 */
        ora_level *L = &levels[level];
        memcpy(L->old_variables, L->variables, sizeof(double) * (size_t)L->nel * ORA_NVAR); /* :383 */
        if (mesh_variant == ORA_MESH_FVCORR)                        /* :388-395 */
            ora_compute_step_factor_legacy(L->nel, L->variables, L->volumes, L->step_factors);
        else
            ora_compute_step_factor(L->nel, L->variables, L->volumes, L->step_factors);
        if (iters) iters[level].compute_step += L->nel;

        for (int j = 0; j < ORA_RK && !rc; j++) {                   /* :397-506 */
            ora_compute_flux_edge(L->internal_start, L->n_internal, L->edges, L->variables, L->fluxes);
            if (iters) iters[level].flux += L->n_internal;
            ora_compute_boundary_flux_edge(L->boundary_start, L->n_boundary, L->edges, L->variables, L->fluxes);
            ora_compute_wall_flux_edge(L->wall_start, L->n_wall, L->edges, L->variables, L->fluxes, &ff);
            ora_time_step(j, L->nel, L->step_factors, L->fluxes, L->old_variables, L->variables);
            if (iters) iters[level].time_step += L->nel;
            rc = ora_check_for_invalid_variables(L->variables, L->nel, NULL);
            if (run_indirect_rw) {
                ora_indirect_rw(L->internal_start, L->n_internal, L->edges, L->variables, L->fluxes);
                ora_zero_fluxes(L->nel, L->fluxes);
            }
            if (iters) iters[level].indirect_rw += L->n_internal;
        }
        if (rc) break;
        ora_residual(L->nel, L->old_variables, L->variables, L->residuals);   /* :508 */
        if (level == 0 && rms_out) rms_out[cyc] = ora_calc_rms(L->nel, L->residuals);

        if (nlevels <= 1) { cyc++; continue; }
        if (going_up) {                                             /* :527-559 */
            level++;
            ora_mg_restrict(levels[level - 1].variables, levels[level].variables, levels[level].nel,
                            levels[level - 1].mg_map, up_scratch, levels[level - 1].mgc);
            /* timer/iteration quirk: booked to the coarse level */
            if (iters) iters[level].restrict_ += 2 * levels[level - 1].mgc + levels[level].nel;
            if (level == nlevels - 1) going_up = 0;
        } else {                                                    /* :560-688 */
            level--;
            ora_prolong_residuals_interpolate_proper(levels[level].edges, levels[level].n_internal,
                    levels[level + 1].residuals, levels[level].residuals, levels[level].variables,
                    levels[level].nel, levels[level].mg_map,
                    levels[level + 1].coords, levels[level].coords);
            if (iters) iters[level].prolong += levels[level].n_internal + levels[level].nel;
            if (level == 0) { going_up = 1; cyc++; }
        }
    }
    free(up_scratch);
    return rc;
}
