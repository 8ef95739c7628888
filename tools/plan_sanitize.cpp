// Host-only: the gather plans of a synthetic two-level hierarchy built under a sanitizer (no device, no HIP):
//   g++ -O1 -g -std=c++17 -fsanitize=thread            -Iinclude -Img-cfd-app-plain_amd/csrc tools/plan_sanitize.cpp mg-cfd-app-plain_amd/csrc/preprocess.cpp -lpthread -o /tmp/plan_tsan && /tmp/plan_tsan
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -Iinclude -Img-cfd-app-plain_amd/csrc tools/plan_sanitize.cpp mg-cfd-app-plain_amd/csrc/preprocess.cpp -lpthread -o /tmp/plan_asan && /tmp/plan_asan
// A jittered n^3 lattice with a few long rows (a hub), its coarse level, the level plan on several host threads (rows filled, tiles
// sorted, the per-tile part) and the transfer plan; then the audit.  argv[1] = n (default 40), MGCFD_PLAN_THREADS as in the library.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#include "mgcfd.h"
#include "preprocess.hpp"
using namespace mgcfd;
struct Level { std::vector<double> vol, xyz; std::vector<mgcfd_edge> edges; std::vector<int64_t> map; mgcfd_level_desc d{}; };
static Level lattice(int n, unsigned seed, bool hub)
{
    Level L; std::mt19937 rng(seed); std::uniform_real_distribution<double> u(-0.2, 0.2);
    const int64_t N = int64_t(n) * n * n;
    auto id = [&](int i, int j, int k) { return (int64_t(i) * n + j) * n + k; };
    L.vol.assign(N, 1.0); L.xyz.resize(3 * N);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) for (int k = 0; k < n; k++) {
        const int64_t a = id(i, j, k); L.xyz[3 * a] = i + u(rng); L.xyz[3 * a + 1] = j + u(rng); L.xyz[3 * a + 2] = k + u(rng); L.vol[a] = 1.0 + 0.1 * u(rng);
    }
    std::vector<mgcfd_edge> in, bnd, wall;
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) for (int k = 0; k < n; k++) {
        const int64_t a = id(i, j, k);
        if (i + 1 < n) in.push_back({a, id(i + 1, j, k), 1.0 + u(rng), u(rng), u(rng)});
        if (j + 1 < n) in.push_back({a, id(i, j + 1, k), u(rng), 1.0 + u(rng), u(rng)});
        if (k + 1 < n) in.push_back({a, id(i, j, k + 1), u(rng), u(rng), 1.0 + u(rng)});
        if (i == 0) wall.push_back({-2, a, -1.0, 0.0, 0.0});
        if (k == 0) bnd.push_back({-1, a, 0.0, 0.0, -1.0});
    }
    if (hub) for (int64_t b = 1; b < std::min<int64_t>(N, 400); b += 3) in.push_back({0, b * 7 % N == 0 ? 1 : b * 7 % N, u(rng), u(rng), u(rng)});   // node 0: a long row
    L.d.nel = N; L.d.internal_start = 0; L.d.n_internal = in.size();
    L.d.boundary_start = in.size(); L.d.n_boundary = bnd.size(); L.d.wall_start = in.size() + bnd.size(); L.d.n_wall = wall.size();
    L.edges = in; L.edges.insert(L.edges.end(), bnd.begin(), bnd.end()); L.edges.insert(L.edges.end(), wall.begin(), wall.end());
    L.d.n_edges = L.edges.size();
    return L;
}
int main(int argc, char **argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 40, nc = n / 2;
    Level F = lattice(n, 1, true), C = lattice(nc, 2, false);
    F.map.resize(F.d.nel);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) for (int k = 0; k < n; k++)
        F.map[(int64_t(i) * n + j) * n + k] = (int64_t(std::min(i / 2, nc - 1)) * nc + std::min(j / 2, nc - 1)) * nc + std::min(k / 2, nc - 1);
    for (Level *L : {&F, &C}) { L->d.volumes = L->vol.data(); L->d.coords = L->xyz.data(); L->d.edges = L->edges.data(); }
    F.d.mg_map = F.map.data(); F.d.mgc = F.d.nel;
    LevelPlan PF, PC; PlanOptions opt;
    for (int order = 0; order < 4; order++) {
        opt.tile_order = order;
        opt.tile_curve = order % 3;                      // (as clustered, Morton, Hilbert)
        PF = LevelPlan(); PC = LevelPlan();
        build_level_plan(F.d, F.edges, opt, PF);
        build_level_plan(C.d, C.edges, opt, PC);
        build_transfer_plan(F.d, F.edges, C.d.coords, C.d.nel, PC.new_of_old, PF, nullptr, F.d.nel);
        const std::string r = audit_level_plan(F.d, PF, C.d.nel) + audit_level_plan(C.d, PC, -1);
        std::printf("tile_order %d: %ld nodes, %d tiles, digest %016llx, audit %s\n", order, (long)F.d.nel, PF.n_tiles, (unsigned long long)plan_digest(PF), r.empty() ? "clean" : r.c_str());
        if (!r.empty()) return 1;
    }
    return 0;
}
