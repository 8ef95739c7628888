// Host-only diagnostic: build the gather plan of level 0 of an input.dat and print what the half-row plan looks like.
//   g++ -O2 -std=c++17 -Iinclude -Img-cfd-app-plain_amd/csrc tools/plan_stats.cpp -Lmg-cfd-app-plain_amd/csrc -lmgcfd_hip -Wl,-rpath,$PWD/mg-cfd-app-plain_amd/csrc -o /tmp/plan_stats
#include <cstdio>
#include <string>
#include <vector>
#include <algorithm>
#include "mgcfd.h"
#include "preprocess.hpp"
int main(int argc, char **argv)
{
    mgcfd_mesh *m = nullptr;
    if (mgcfd_mesh_load("input.dat", argv[1], 1, &m) != MGCFD_OK) { std::fprintf(stderr, "%s\n", mgcfd_last_error()); return 1; }
    mgcfd_level_desc d;
    mgcfd_mesh_level(m, 0, &d);
    std::vector<mgcfd_edge> edges(d.edges, d.edges + d.n_edges);
    mgcfd::adjust_and_dampen(d, mgcfd_mesh_variant(m), edges);
    mgcfd::LevelPlan P;
    mgcfd::PlanOptions opt;
    mgcfd::build_level_plan(d, edges, opt, P);
    std::printf("nel %ld tiles %d halo mean %.1f max %d overflow refs %ld tail %d edge_once %d\n", (long)d.nel, P.n_tiles, P.halo_mean, P.halo_max,
                (long)P.halo_overflow_refs, int(P.has_tail), int(P.edge_once));
    std::printf("edges listed per tile: mean %.1f max %d (24 B each: %.1f KB at most)\n", P.te_mean, P.te_max, P.te_max * 24 / 1024.0);
    std::printf("half %d evaluations %ld (%.2f per node) foreign %ld padding %ld (%.1f %%)\n", int(P.half), (long)P.hr_entries, double(P.hr_entries) / d.nel,
                (long)P.hr_foreign, (long)P.hr_padding, P.hr_entries ? 100.0 * P.hr_padding / P.hr_entries : 0.0);
    {   // how scattered the halo gathers are: distinct 128-byte blocks (16 consecutive ids) per tile, per field
        int64_t blocks = 0, nodes = 0;
        for (int32_t t = 0; t < P.n_tiles; t++) {
            int32_t last = -1;
            for (int32_t k = P.tile_halo_ptr[t]; k < P.tile_halo_ptr[t + 1]; k++) {
                const int32_t b = P.tile_halo[k] / 16;
                if (b != last) { blocks++; last = b; }
                nodes++;
            }
        }
        std::printf("halo gathers: %.1f nodes per tile in %.1f blocks of 16 ids (%.2f nodes per block)\n", double(nodes) / P.n_tiles, double(blocks) / P.n_tiles, double(nodes) / blocks);
    }
    {   // every index the kernels form from this plan against the size of what it indexes (preprocess.cpp)
        const std::string rep = mgcfd::audit_level_plan(d, P, -1);
        std::printf("bounds audit: %s\n", rep.empty() ? "every index in range (LDS slots, halo / overflow positions, half-row owners, list entries; role-5 precondition)" : rep.c_str());
        const long stride = long(P.n_slices) * 64;
        std::printf("  arrays a stage launch indexes by node: stride %ld (state, fluxes, residuals 5 x stride; volumes, step factors stride); largest node index formed %ld; "
                    "tile_halo %d x %d ids, largest position %d; step-factor partials %d, index clamped to %d\n", stride, stride - 1, P.n_tiles, mgcfd::kHaloStride,
                    (P.n_tiles - 1) * mgcfd::kHaloStride + mgcfd::kHaloStride - 1, P.n_tiles, P.n_tiles - 1);
    }
    if (!P.hr_row0.empty()) {
        int hist[16] = {0};
        for (size_t s = 0; s + 1 < P.hr_row0.size(); s++) hist[std::min(15, P.hr_row0[s + 1] - P.hr_row0[s])]++;
        for (int k = 0; k < 16; k++) if (hist[k]) std::printf("  slices with %d half rows: %d\n", k, hist[k]);
        // where the evaluations a lane does for ANOTHER node sit (the order-free kernel takes a slower path for a row pair that holds one)
        long rows = 0, rows_f = 0, pairs = 0, pairs_f = 0, by_pos[16] = {0}, lanes_by_pos[16] = {0};
        for (size_t s = 0; s + 1 < P.hr_row0.size(); s++) {
            bool prev = false;
            for (int32_t r = P.hr_row0[s]; r < P.hr_row0[s + 1]; r++) {
                int n = 0;
                for (int l = 0; l < 64; l++) n += (P.hr_code[size_t(r) * 64 + l] & mgcfd::kHalfForeign) != 0;
                const int pos = std::min(15, int(r - P.hr_row0[s]));
                rows++; rows_f += n > 0; by_pos[pos] += n > 0; lanes_by_pos[pos] += n;
                if (pos % 2 == 0) { pairs++; prev = n > 0; if (r + 1 == P.hr_row0[s + 1]) pairs_f += prev; }
                else pairs_f += prev || n > 0;
            }
        }
        std::printf("  half rows holding a foreign evaluation: %ld of %ld; row pairs: %ld of %ld\n", rows_f, rows, pairs_f, pairs);
        for (int k = 0; k < 16; k++) if (by_pos[k]) std::printf("    row %d of its slice: %ld rows, %ld lanes\n", k, by_pos[k], lanes_by_pos[k]);
    }
    return 0;
}
