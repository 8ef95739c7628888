// preprocess.cpp — renumbering and gather-structure construction (host, once per mesh).
// Compiled with -ffp-contract=off: the static weights computed here (edge length factor,
// inverse distances, weight sums) must carry the same bits the reference computes at run time.
#include "preprocess.hpp"

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <numeric>
#include <deque>
#include <limits>
#include <queue>
#include <stdexcept>
#include <string>
#include <thread>

namespace mgcfd {

namespace {

// src/Base/common.h:24 — a float literal widened to double.
const double kSmoothing = double(0.2f);

struct Adjacency {
    std::vector<int32_t> ptr, idx;
};

Adjacency build_adjacency(int64_t nel, const mgcfd_edge *edges, int64_t first, int64_t count)
{
    Adjacency g;
    g.ptr.assign(static_cast<size_t>(nel) + 1, 0);
    for (int64_t e = first; e < first + count; e++) {
        g.ptr[static_cast<size_t>(edges[e].a) + 1]++;
        g.ptr[static_cast<size_t>(edges[e].b) + 1]++;
    }
    std::partial_sum(g.ptr.begin(), g.ptr.end(), g.ptr.begin());
    g.idx.resize(static_cast<size_t>(g.ptr.back()));
    std::vector<int32_t> fill(g.ptr.begin(), g.ptr.end() - 1);
    for (int64_t e = first; e < first + count; e++) {
        const int32_t a = static_cast<int32_t>(edges[e].a), b = static_cast<int32_t>(edges[e].b);
        g.idx[static_cast<size_t>(fill[a]++)] = b;
        g.idx[static_cast<size_t>(fill[b]++)] = a;
    }
    return g;
}

// Breadth-first (Cuthill–McKee) ordering: a wavefront's 64 consecutive nodes then have
// their r-th neighbours in a narrow, mostly ascending window of the node array, which is
// what makes the per-row gathers cache friendly.  Handles disconnected meshes (-m copies).
std::vector<int32_t> cuthill_mckee(const Adjacency &g, int64_t nel)
{
    std::vector<int32_t> order;
    order.reserve(static_cast<size_t>(nel));
    std::vector<char> seen(static_cast<size_t>(nel), 0);
    auto degree = [&](int32_t v) { return g.ptr[static_cast<size_t>(v) + 1] - g.ptr[static_cast<size_t>(v)]; };
    std::vector<int32_t> level_nodes, scratch;

    auto bfs_far_node = [&](int32_t start) {
        // returns the last node reached from `start` (an approximately peripheral node)
        std::vector<int32_t> frontier{start}, next;
        std::vector<int32_t> touched{start};
        seen[static_cast<size_t>(start)] = 2;
        int32_t last = start;
        while (!frontier.empty()) {
            next.clear();
            for (int32_t v : frontier)
                for (int32_t k = g.ptr[static_cast<size_t>(v)]; k < g.ptr[static_cast<size_t>(v) + 1]; k++) {
                    int32_t u = g.idx[static_cast<size_t>(k)];
                    if (!seen[static_cast<size_t>(u)]) { seen[static_cast<size_t>(u)] = 2; next.push_back(u); touched.push_back(u); }
                }
            if (!next.empty()) {
                last = *std::min_element(next.begin(), next.end(), [&](int32_t x, int32_t y) { return degree(x) < degree(y); });
            }
            frontier.swap(next);
        }
        for (int32_t v : touched) seen[static_cast<size_t>(v)] = 0;
        return last;
    };

    for (int64_t s0 = 0; s0 < nel; s0++) {
        if (seen[static_cast<size_t>(s0)]) continue;
        int32_t start = bfs_far_node(static_cast<int32_t>(s0));
        start = bfs_far_node(start);
        size_t head = order.size();
        order.push_back(start);
        seen[static_cast<size_t>(start)] = 1;
        while (head < order.size()) {
            int32_t v = order[head++];
            scratch.clear();
            for (int32_t k = g.ptr[static_cast<size_t>(v)]; k < g.ptr[static_cast<size_t>(v) + 1]; k++) {
                int32_t u = g.idx[static_cast<size_t>(k)];
                if (!seen[static_cast<size_t>(u)]) { seen[static_cast<size_t>(u)] = 1; scratch.push_back(u); }
            }
            std::sort(scratch.begin(), scratch.end(), [&](int32_t x, int32_t y) {
                int dx = degree(x), dy = degree(y);
                return dx != dy ? dx < dy : x < y;
            });
            order.insert(order.end(), scratch.begin(), scratch.end());
        }
    }
    return order;   // order[new] = old
}

// Greedy graph-growing partition into compact clusters of kTile nodes: grow a breadth-first
// ball from a seed until it holds kTile nodes; the nodes left on its frontier seed the next
// clusters.  Consecutive chunks of kTile nodes of the returned order are the clusters.  A ball
// of 256 nodes in a 3-D mesh has a few hundred neighbours outside it, against ~600 for 256
// consecutive nodes of a breadth-first band — that halo is what a tile stages in LDS.
std::vector<int32_t> cluster_order(const Adjacency &g, int64_t nel)
{
    std::vector<int32_t> order;
    order.reserve(static_cast<size_t>(nel));
    std::vector<char> state(static_cast<size_t>(nel), 0);      // 0 free, 1 taken, 2 queued in the current ball
    std::deque<int32_t> seeds;
    std::vector<int32_t> ball;
    int64_t scan = 0;
    while (static_cast<int64_t>(order.size()) < nel) {
        int32_t seed = -1;
        while (!seeds.empty()) {
            int32_t c = seeds.front();
            seeds.pop_front();
            if (state[static_cast<size_t>(c)] == 0) { seed = c; break; }
        }
        if (seed < 0) {
            while (state[static_cast<size_t>(scan)] != 0) scan++;
            seed = static_cast<int32_t>(scan);
        }
        // fill one tile; if a component runs out first, continue from the next seed so every
        // tile (but the last) holds exactly kTile nodes
        const size_t tile_end = std::min<size_t>(static_cast<size_t>(nel), (order.size() / kTile + 1) * kTile);
        ball.clear();
        ball.push_back(seed);
        state[static_cast<size_t>(seed)] = 2;
        size_t head = 0;
        while (head < ball.size() && order.size() < tile_end) {
            const int32_t v = ball[head++];
            state[static_cast<size_t>(v)] = 1;
            order.push_back(v);
            for (int32_t k = g.ptr[static_cast<size_t>(v)]; k < g.ptr[static_cast<size_t>(v) + 1]; k++) {
                const int32_t u = g.idx[static_cast<size_t>(k)];
                if (state[static_cast<size_t>(u)] == 0) { state[static_cast<size_t>(u)] = 2; ball.push_back(u); }
            }
        }
        for (; head < ball.size(); head++) {                   // frontier left over: future seeds
            state[static_cast<size_t>(ball[head])] = 0;
            seeds.push_back(ball[head]);
        }
    }
    return order;
}

// Recursive coordinate bisection into boxes of kTile nodes (all but the last hold exactly kTile): split the node
// set at a multiple of kTile along its longest axis.  No fragments, whatever the degree distribution — the fallback
// for meshes (tetrahedral, ~15 neighbours per node) on which the greedy balls leave ragged remainders between them.
void rcb_split(const double *coords, int32_t *ids, int64_t n)
{
    if (n <= kTile) return;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int64_t i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) {
            const double x = coords[3 * static_cast<size_t>(ids[i]) + c];
            lo[c] = std::min(lo[c], x); hi[c] = std::max(hi[c], x);
        }
    int ax = 0;
    for (int c = 1; c < 3; c++) if (hi[c] - lo[c] > hi[ax] - lo[ax]) ax = c;
    const int64_t tiles = (n + kTile - 1) / kTile, left = (tiles / 2) * kTile;
    std::nth_element(ids, ids + left, ids + n, [&](int32_t x, int32_t y) {
        const double cx = coords[3 * static_cast<size_t>(x) + ax], cy = coords[3 * static_cast<size_t>(y) + ax];
        return cx != cy ? cx < cy : x < y;
    });
    rcb_split(coords, ids, left);
    rcb_split(coords, ids + left, n - left);
}

// What an ordering costs the tile kernels: {halo nodes that do not fit a tile's LDS (read from HBM per use), all halo nodes}.
std::pair<int64_t, int64_t> halo_cost(const Adjacency &g, const std::vector<int32_t> &order, int64_t nel)
{
    std::vector<int32_t> tile_of(static_cast<size_t>(nel)), stamp(static_cast<size_t>(nel), -1);
    for (int64_t n = 0; n < nel; n++) tile_of[static_cast<size_t>(order[static_cast<size_t>(n)])] = static_cast<int32_t>(n / kTile);
    int64_t excess = 0, total = 0;
    for (int64_t s = 0; s < nel; s += kTile) {
        const int32_t t = static_cast<int32_t>(s / kTile);
        int64_t count = 0;
        for (int64_t n = s; n < std::min(nel, s + kTile); n++) {
            const int32_t v = order[static_cast<size_t>(n)];
            for (int32_t k = g.ptr[static_cast<size_t>(v)]; k < g.ptr[static_cast<size_t>(v) + 1]; k++) {
                const int32_t u = g.idx[static_cast<size_t>(k)];
                if (tile_of[static_cast<size_t>(u)] != t && stamp[static_cast<size_t>(u)] != t) { stamp[static_cast<size_t>(u)] = t; count++; }
            }
        }
        total += count;
        excess += std::max<int64_t>(0, count - kHaloStride);
    }
    return {excess, total};
}

inline double inv_distance(const double *p, const double *q)
{
    const double dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
    return 1.0 / std::sqrt(dx * dx + dy * dy + dz * dz);
}

inline bool coincident(const double *fine, const double *coarse)
{
    return (fine[0] - coarse[0]) == 0.0 && (fine[1] - coarse[1]) == 0.0 && (fine[2] - coarse[2]) == 0.0;
}

} // namespace

void far_field_constants(double *out)
{
    // src/Kernels/cfd_loops.h:85-119, src/Base/const.h:9-15
    const double gamma = 1.4, ff_mach = 1.2, deg_aoa = 0.0;
    const double angle = double(3.1415926535897931 / 180.0) * double(deg_aoa);
    double var[5];
    var[0] = 1.4;
    const double pressure = 1.0;
    const double c = std::sqrt(gamma * pressure / var[0]);
    const double speed = ff_mach * c;
    const double vx = speed * std::cos(angle), vy = speed * std::sin(angle), vz = 0.0;
    var[1] = var[0] * vx;
    var[2] = var[0] * vy;
    var[3] = var[0] * vz;
    var[4] = var[0] * (0.5 * (speed * speed)) + (pressure / (gamma - 1.0));
    for (int v = 0; v < 5; v++) out[v] = var[v];
    // compute_flux_contribution, cfd_loops.h:57-83
    double *mx = out + 5, *my = out + 8, *mz = out + 11, *de = out + 14;
    mx[0] = vx * var[1] + pressure; mx[1] = vx * var[2]; mx[2] = vx * var[3];
    my[0] = mx[1]; my[1] = vy * var[2] + pressure; my[2] = vy * var[3];
    mz[0] = mx[2]; mz[1] = my[2]; mz[2] = vz * var[3] + pressure;
    const double de_p = var[4] + pressure;
    de[0] = vx * de_p; de[1] = vy * de_p; de[2] = vz * de_p;
}

void adjust_and_dampen(const mgcfd_level_desc &L, int mesh_variant, std::vector<mgcfd_edge> &edges)
{
    double damping = 0.0;
    if (mesh_variant == MGCFD_MESH_M6_WING) damping = 5e-8;
    else if (mesh_variant == MGCFD_MESH_LA_CASCADE) damping = 1e-7;
    else if (mesh_variant == MGCFD_MESH_ROTOR_37) damping = 2e-7;
    if (damping == 0.0) return;
    if (!L.coords) throw std::runtime_error("mesh variant needs node coordinates (.coords) for adjust_ewt");
    for (auto &e : edges) {
        if (e.a >= 0 && e.b >= 0) {
            // |w| is a face area; divide by the node distance (validation.cpp:41-55)
            const double *ca = L.coords + 3 * e.a, *cb = L.coords + 3 * e.b;
            double dist = 0.0, d;
            d = cb[0] - ca[0]; dist += d * d;
            d = cb[1] - ca[1]; dist += d * d;
            d = cb[2] - ca[2]; dist += d * d;
            dist = std::sqrt(dist);
            e.x /= dist; e.y /= dist; e.z /= dist;
        }
        e.x *= damping; e.y *= damping; e.z *= damping;       // validation.cpp:70-74
    }
}

void build_level_plan(const mgcfd_level_desc &L, const std::vector<mgcfd_edge> &edges,
                      const PlanOptions &opt, LevelPlan &P)
{
    // MGCFD_PLAN_TIMING=1: where the host time of a level's plan goes (stderr)
    const bool timing = std::getenv("MGCFD_PLAN_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[mgcfd plan %ld nodes] %-28s %7.1f ms\n", (long)L.nel, what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    const int64_t nel = L.nel;
    if (nel <= 0 || nel >= (int64_t(1) << 29)) throw std::runtime_error("level size out of range for 29-bit node ids");
    // host threads for the parts of the plan that are independent per node or per tile (the result never depends on their number)
    int plan_threads = opt.threads > 0 ? opt.threads : int(std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency())));
    plan_threads = std::max<int>(1, std::min<int64_t>(plan_threads, nel / (64 * kTile)));
    if (const char *o = std::getenv("MGCFD_PLAN_THREADS")) plan_threads = std::max<int>(1, std::min<int64_t>(std::atoi(o), std::max<int64_t>(1, nel / kTile)));
    auto over_ranges = [&](int64_t n, int64_t granule, auto &&body) {        // body(begin, end) on [0, n) cut at multiples of `granule`
        const int64_t units = (n + granule - 1) / granule;
        const int k_max = static_cast<int>(std::min<int64_t>(plan_threads, std::max<int64_t>(1, units)));
        std::vector<std::exception_ptr> errors(static_cast<size_t>(k_max));
        auto guarded = [&](int k) {
            try { body(std::min(n, units * k / k_max * granule), std::min(n, units * (k + 1) / k_max * granule)); }
            catch (...) { errors[static_cast<size_t>(k)] = std::current_exception(); }
        };
        std::vector<std::thread> workers;
        for (int k = 1; k < k_max; k++) workers.emplace_back(guarded, k);
        guarded(0);
        for (std::thread &w : workers) w.join();
        for (const std::exception_ptr &e : errors) if (e) std::rethrow_exception(e);
    };
    for (int64_t e = L.internal_start; e < L.internal_start + L.n_internal; e++)
        if (edges[e].a < 0 || edges[e].a >= nel || edges[e].b < 0 || edges[e].b >= nel)
            throw std::runtime_error("internal edge " + std::to_string(e) + " has an end point outside [0, nel)");
    for (int64_t e = L.boundary_start; e < L.boundary_start + L.n_boundary; e++)
        if (edges[e].b < 0 || edges[e].b >= nel) throw std::runtime_error("boundary edge with bad node");
    for (int64_t e = L.wall_start; e < L.wall_start + L.n_wall; e++)
        if (edges[e].b < 0 || edges[e].b >= nel) throw std::runtime_error("wall edge with bad node");

    P.nel = nel;
    const Adjacency g = build_adjacency(nel, edges.data(), L.internal_start, L.n_internal);
    const int64_t n_owned = (opt.n_owned < 0 || opt.n_owned > nel) ? nel : opt.n_owned;
    auto owned = [&](int64_t original_id) { return original_id < n_owned; };
    // incidence rows a node will actually hold (ghosts hold none)
    auto rows_of = [&](int32_t v) -> int32_t {
        return owned(v) ? g.ptr[static_cast<size_t>(v) + 1] - g.ptr[static_cast<size_t>(v)] : 0;
    };

    lap("adjacency");
    // ---- node order ----
    // A partitioned level (n_owned < nel): the OWNED nodes are clustered among themselves and numbered first, the ghosts
    // follow in their given order.  Tiles past the owned nodes hold only ghosts and are never launched, and a launch that
    // is told nel = n_owned leaves every ghost slot alone: ghosts are written by halo messages only (solver.cpp: a peer
    // may store into them while this rank's own kernels run).
    std::vector<int32_t> order;
    if (opt.ordering == 2 && n_owned < nel) {
        std::vector<mgcfd_edge> inner;
        for (int64_t e = L.internal_start; e < L.internal_start + L.n_internal; e++)
            if (owned(edges[static_cast<size_t>(e)].a) && owned(edges[static_cast<size_t>(e)].b)) inner.push_back(edges[static_cast<size_t>(e)]);
        const Adjacency go = build_adjacency(n_owned, inner.data(), 0, static_cast<int64_t>(inner.size()));
        order = cluster_order(go, n_owned);
        auto with_ghosts = [&](std::vector<int32_t> v) {
            for (int64_t v2 = n_owned; v2 < nel; v2++) v.push_back(static_cast<int32_t>(v2));
            return v;
        };
        order = with_ghosts(order);
        P.ghosts_last = true;
        const std::pair<int64_t, int64_t> greedy = halo_cost(g, order, nel);
        if (greedy.first > 0 && L.coords) {
            std::vector<int32_t> boxes(static_cast<size_t>(n_owned));
            std::iota(boxes.begin(), boxes.end(), 0);
            rcb_split(L.coords, boxes.data(), n_owned);
            boxes = with_ghosts(boxes);
            if (halo_cost(g, boxes, nel) < greedy) { order.swap(boxes); P.ordered_by_boxes = true; }
        }
    }
    else if (opt.ordering == 2) {
        order = cluster_order(g, nel);
        // greedy balls have the smallest halos on hexahedral-like meshes; where they overflow the LDS tile
        // (high-degree meshes) try coordinate boxes and keep whichever leaves fewer halo nodes outside
        const std::pair<int64_t, int64_t> greedy = halo_cost(g, order, nel);
        if (greedy.first > 0 && L.coords) {
            std::vector<int32_t> boxes(static_cast<size_t>(nel));
            std::iota(boxes.begin(), boxes.end(), 0);
            rcb_split(L.coords, boxes.data(), nel);
            if (halo_cost(g, boxes, nel) < greedy) { order.swap(boxes); P.ordered_by_boxes = true; }
        }
    }
    else if (opt.ordering == 1) order = cuthill_mckee(g, nel);
    else { order.resize(static_cast<size_t>(nel)); std::iota(order.begin(), order.end(), 0); }
    // ---- dispatch order of the tiles ----
    // A launch deals its workgroups to the XCDs round-robin and gives every XCD one contiguous range of tiles
    // (kernels.hip: xcd_contiguous_block), so within a range the tiles start in index order.  A level of more tiles than
    // the chip holds workgroups (768 of the bit-identical kernel: three per CU) runs in 1.15-1.5 rounds, and what starts
    // last decides when the launch ends.  So the cheapest tiles of every range — as many as do not fit the first round —
    // go to its end, the costliest of them first; every other tile keeps its place (neighbours in space stay neighbours in
    // time: their halos meet in the XCD's L2).  Cost: the longest row a lane of the tile walks, then the nodes it stages
    // beside its own.  Measured on the 67^3 mixed-element level (flux launch): 19.2 -> 18.2 us bit-identical, 14.9 -> 14.2
    // order-free; sorting the WHOLE range by cost (tile_order = 1) gains the same there and costs a lattice's bit-identical
    // stages 1 %.  Any order gives the same results.
    // First, the complete owned tiles along a space-filling curve through their centroids (Morton order; tile_curve = 2: Hilbert).
    // The clustering grows its tiles breadth-first, so a run of consecutive tiles — an XCD's range — is a thin shell across the
    // whole mesh, and a tile's neighbours in the layers before and behind it lie a hundred tiles away: in another XCD's range, or
    // long gone from this one's L2.  The counters showed it: of the 8.9 MB of halo state a launch on the 67^3 lattice reads, 8.3 MB
    // came from the fabric.  Along the curve an XCD's range is a compact block and consecutive tiles are neighbours: flux launch
    // 15.87 -> 15.59 us bit-identical and 13.90 -> 13.16 order-free, sweeps 54.4 -> 53.0 / 46.9 -> 46.0 us, V-cycle 0.2875 ->
    // 0.279 / 0.258 -> 0.2485 ms (same box, two runs each; the mixed-element level: unchanged).  Needs coordinates.
    if (opt.ordering == 2 && L.coords && opt.tile_curve != 0) {
        const int64_t n_perm = n_owned / kTile;
        if (n_perm > 8) {
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            std::vector<double> c(static_cast<size_t>(n_perm) * 3, 0.0);
            for (int64_t t = 0; t < n_perm; t++) {
                for (int64_t n = t * kTile; n < (t + 1) * kTile; n++)
                    for (int d = 0; d < 3; d++) c[static_cast<size_t>(t) * 3 + d] += L.coords[static_cast<size_t>(order[static_cast<size_t>(n)]) * 3 + d];
                for (int d = 0; d < 3; d++) { c[static_cast<size_t>(t) * 3 + d] /= kTile; lo[d] = std::min(lo[d], c[static_cast<size_t>(t) * 3 + d]); hi[d] = std::max(hi[d], c[static_cast<size_t>(t) * 3 + d]); }
            }
            auto spread = [](uint64_t v) { uint64_t r = 0; for (int b = 0; b < 16; b++) r |= ((v >> b) & 1ull) << (3 * b); return r; };
            const bool hilbert = opt.tile_curve == 2;
            std::vector<std::pair<uint64_t, int64_t>> key(static_cast<size_t>(n_perm));
            for (int64_t t = 0; t < n_perm; t++) {
                uint32_t X[3];
                for (int d = 0; d < 3; d++) {
                    double f = hi[d] > lo[d] ? (c[static_cast<size_t>(t) * 3 + d] - lo[d]) / (hi[d] - lo[d]) : 0.0;
                    if (!(f >= 0.0)) f = 0.0;                             // (a coordinate that is not a number: any place on the curve will do)
                    if (f > 1.0) f = 1.0;
                    X[d] = static_cast<uint32_t>(f * 65535.0);
                }
                if (hilbert) {
                    // axes -> transposed Hilbert index (J. Skilling, "Programming the Hilbert curve", 2004), 16 bits per axis
                    const uint32_t M = 1u << 15;
                    for (uint32_t Q = M; Q > 1; Q >>= 1) {
                        const uint32_t P2 = Q - 1;
                        for (int d = 0; d < 3; d++) {
                            if (X[d] & Q) X[0] ^= P2;
                            else { const uint32_t tt = (X[0] ^ X[d]) & P2; X[0] ^= tt; X[d] ^= tt; }
                        }
                    }
                    for (int d = 1; d < 3; d++) X[d] ^= X[d - 1];
                    uint32_t tt = 0;
                    for (uint32_t Q = M; Q > 1; Q >>= 1) if (X[2] & Q) tt ^= Q - 1;
                    for (int d = 0; d < 3; d++) X[d] ^= tt;
                    // (transposed: bit b of the index's digit comes from X[0], X[1], X[2] in turn, most significant first)
                    key[static_cast<size_t>(t)] = {(spread(X[0]) << 2) | (spread(X[1]) << 1) | spread(X[2]), t};
                } else
                    key[static_cast<size_t>(t)] = {spread(X[0]) | (spread(X[1]) << 1) | (spread(X[2]) << 2), t};
            }
            std::sort(key.begin(), key.end());
            std::vector<int32_t> reordered(order);
            for (int64_t k = 0; k < n_perm; k++)
                std::copy(order.begin() + key[static_cast<size_t>(k)].second * kTile, order.begin() + (key[static_cast<size_t>(k)].second + 1) * kTile, reordered.begin() + k * kTile);
            order.swap(reordered);
        }
    }
    if (opt.ordering == 2 && opt.tile_order != 0) {
        const int64_t n_tiles_all = (nel + kTile - 1) / kTile;
        const int64_t n_perm = n_owned / kTile;                          // complete tiles of owned nodes only
        const int64_t first_round = 768;
        if (n_perm > 8 && (opt.tile_order == 1 || n_tiles_all > first_round)) {
            std::vector<int64_t> cost(static_cast<size_t>(n_perm), 0);
            int64_t lo_rows = std::numeric_limits<int64_t>::max(), hi_rows = 0;
            for (int64_t t = 0; t < n_perm; t++) {
                int32_t longest = 0;
                for (int64_t n = t * kTile; n < (t + 1) * kTile; n++) longest = std::max(longest, rows_of(order[static_cast<size_t>(n)]));
                cost[static_cast<size_t>(t)] = int64_t(longest) * 1024;
                lo_rows = std::min<int64_t>(lo_rows, longest); hi_rows = std::max<int64_t>(hi_rows, longest);
            }
            // (the halos — a walk over every edge — only where an order will be made of them)
            if (opt.tile_order == 1 || opt.tile_order == 3 || hi_rows >= lo_rows + 2 || std::getenv("MGCFD_VERBOSE")) {
                std::vector<int32_t> tile_of(static_cast<size_t>(nel));
                for (int64_t n = 0; n < nel; n++) tile_of[static_cast<size_t>(order[static_cast<size_t>(n)])] = static_cast<int32_t>(n / kTile);
                std::vector<int32_t> seen(static_cast<size_t>(nel), -1);
                for (int64_t t = 0; t < n_perm; t++) {
                    int32_t halo = 0;
                    for (int64_t n = t * kTile; n < (t + 1) * kTile; n++) {
                        const int32_t v = order[static_cast<size_t>(n)];
                        for (int32_t k = g.ptr[static_cast<size_t>(v)]; k < g.ptr[static_cast<size_t>(v) + 1]; k++) {
                            const int32_t w = g.idx[static_cast<size_t>(k)];
                            if (tile_of[static_cast<size_t>(w)] != t && seen[static_cast<size_t>(w)] != t) { seen[static_cast<size_t>(w)] = static_cast<int32_t>(t); halo++; }
                        }
                    }
                    cost[static_cast<size_t>(t)] += halo;
                }
            }
            // (tiles that all walk rows of the same length — a lattice — differ by their halos only, and moving those costs
            //  the later rounds of a large level more in locality than the tail gains: 84^3 and 96^3 lattices 1-2.5 % slower,
            //  67^3 and 134^3 unchanged; such a level keeps its order)
            const bool rows_differ = hi_rows >= lo_rows + 2;
            std::vector<int32_t> reordered(order);
            const int64_t q = n_tiles_all >> 3, r = n_tiles_all & 7;
            if (std::getenv("MGCFD_VERBOSE")) {
                int64_t f0 = 0;
                for (int64_t x = 0; x < 8; x++) {
                    const int64_t l0 = std::min(f0 + q + (x < r ? 1 : 0), n_perm);
                    int64_t rows_sum = 0, halo_sum = 0;
                    for (int64_t t = f0; t < l0; t++) { rows_sum += cost[static_cast<size_t>(t)] / 1024; halo_sum += cost[static_cast<size_t>(t)] % 1024; }
                    std::fprintf(stderr, "[mgcfd] XCD %ld: tiles %ld..%ld, sum of longest rows %ld, sum of halo nodes %ld\n", (long)x, (long)f0, (long)l0, (long)rows_sum, (long)halo_sum);
                    f0 += q + (x < r ? 1 : 0);
                }
            }
            // (never more than one round's share: in a launch of several rounds only the last one is the tail)
            int64_t n_tail_of_a_range = std::min<int64_t>(std::max<int64_t>(0, (n_tiles_all - first_round + 7) / 8), first_round / 8);
            int64_t first = 0;
            for (int64_t x = 0; x < 8; x++) {
                const int64_t last = std::min(first + q + (x < r ? 1 : 0), n_perm);
                if (last > first) {
                    std::vector<int64_t> ids(static_cast<size_t>(last - first));
                    std::iota(ids.begin(), ids.end(), first);
                    std::vector<int64_t> by_cost(ids);
                    std::stable_sort(by_cost.begin(), by_cost.end(), [&](int64_t a, int64_t b) { return cost[static_cast<size_t>(a)] > cost[static_cast<size_t>(b)]; });
                    if (opt.tile_order == 1) ids.swap(by_cost);
                    else if (rows_differ || opt.tile_order == 3) {
                        const int64_t n_tail = std::min<int64_t>(last - first, n_tail_of_a_range);
                        std::vector<char> in_tail(static_cast<size_t>(last - first), 0);
                        for (int64_t k = (last - first) - n_tail; k < last - first; k++) in_tail[static_cast<size_t>(by_cost[static_cast<size_t>(k)] - first)] = 1;
                        std::vector<int64_t> kept;
                        for (int64_t t : ids) if (!in_tail[static_cast<size_t>(t - first)]) kept.push_back(t);
                        for (int64_t k = (last - first) - n_tail; k < last - first; k++) kept.push_back(by_cost[static_cast<size_t>(k)]);
                        ids.swap(kept);
                    }
                    for (int64_t k = 0; k < last - first; k++)
                        std::copy(order.begin() + ids[static_cast<size_t>(k)] * kTile, order.begin() + (ids[static_cast<size_t>(k)] + 1) * kTile, reordered.begin() + (first + k) * kTile);
                }
                first += q + (x < r ? 1 : 0);
            }
            order.swap(reordered);
        }
    }
    lap("node order");
    // ---- half rows (preprocess.hpp): which end point evaluates an internal edge.  An edge whose end points share a
    //      tile (and are both owned) is evaluated by ONE of them; any other edge by each owned end point in its own tile.
    //      Orientation: greedy to the less loaded end, then flips from a node to a neighbour at least two below it until
    //      none is left (loads of adjacent nodes then differ by at most one along every such edge). ----
    std::vector<int8_t> half_eval_a(static_cast<size_t>(L.n_internal), int8_t(-1));
    std::vector<int32_t> half_load(static_cast<size_t>(nel), 0);
    {
        std::vector<int32_t> tile_of(static_cast<size_t>(nel));
        for (int64_t n = 0; n < nel; n++) tile_of[static_cast<size_t>(order[static_cast<size_t>(n)])] = static_cast<int32_t>(n / kTile);
        std::vector<int64_t> shared;
        for (int64_t k = 0; k < L.n_internal; k++) {
            const mgcfd_edge &E = edges[static_cast<size_t>(L.internal_start + k)];
            if (owned(E.a) && owned(E.b) && tile_of[static_cast<size_t>(E.a)] == tile_of[static_cast<size_t>(E.b)]) shared.push_back(k);
            else { if (owned(E.a)) half_load[static_cast<size_t>(E.a)]++; if (owned(E.b)) half_load[static_cast<size_t>(E.b)]++; }
        }
        for (int64_t k : shared) {
            const mgcfd_edge &E = edges[static_cast<size_t>(L.internal_start + k)];
            const bool to_a = half_load[static_cast<size_t>(E.a)] <= half_load[static_cast<size_t>(E.b)];
            half_eval_a[static_cast<size_t>(k)] = to_a ? 1 : 0;
            half_load[static_cast<size_t>(to_a ? E.a : E.b)]++;
        }
        for (int pass = 0; pass < 64; pass++) {
            bool moved = false;
            for (int64_t k : shared) {
                const mgcfd_edge &E = edges[static_cast<size_t>(L.internal_start + k)];
                const bool at_a = half_eval_a[static_cast<size_t>(k)] == 1;
                int32_t &from = half_load[static_cast<size_t>(at_a ? E.a : E.b)], &to = half_load[static_cast<size_t>(at_a ? E.b : E.a)];
                if (from >= to + 2) { from--; to++; half_eval_a[static_cast<size_t>(k)] = at_a ? 0 : 1; moved = true; }
            }
            if (!moved) break;
        }
    }
    lap("half-row orientation");
    std::vector<int32_t> bnd_count(static_cast<size_t>(nel), 0);
    for (int64_t e = L.boundary_start; e < L.boundary_start + L.n_boundary; e++) if (owned(edges[e].b)) bnd_count[static_cast<size_t>(edges[e].b)]++;
    for (int64_t e = L.wall_start; e < L.wall_start + L.n_wall; e++) if (owned(edges[e].b)) bnd_count[static_cast<size_t>(edges[e].b)]++;
    if (opt.degree_sort) {
        // Inside each tile sort by (internal degree, boundary faces) so the 64 nodes of a
        // slice have equal row counts and ELL padding stays small; tile membership is kept.
        // Third key: the number of edges the node evaluates in the half-row plan (decided above: it depends on which
        // nodes share a tile, not on their order inside it), so the lanes of a slice walk equally many half rows.
        const std::vector<int32_t> &cut_count = half_load;
        // ... and before it: the OTHER tile a node faces (the smallest tile among its outside neighbours; none: last).
        // A tile gathers its halo by id from the tiles around it, five scattered 8-byte loads per halo node; with a
        // tile's nodes grouped by the neighbour they face, the ids one neighbour asks for are runs of consecutive
        // ids — a few 128-byte lines per field instead of one line per node.
        std::vector<int32_t> facing(static_cast<size_t>(nel), std::numeric_limits<int32_t>::max());
        if (!std::getenv("MGCFD_NO_FACING_SORT")) {
            std::vector<int32_t> tile_of(static_cast<size_t>(nel));
            for (int64_t n = 0; n < nel; n++) tile_of[static_cast<size_t>(order[static_cast<size_t>(n)])] = static_cast<int32_t>(n / kTile);
            for (int64_t v = 0; v < nel; v++)
                for (int32_t k = g.ptr[static_cast<size_t>(v)]; k < g.ptr[static_cast<size_t>(v) + 1]; k++) {
                    const int32_t tw = tile_of[static_cast<size_t>(g.idx[static_cast<size_t>(k)])];
                    if (tw != tile_of[static_cast<size_t>(v)]) facing[static_cast<size_t>(v)] = std::min(facing[static_cast<size_t>(v)], tw);
                }
        }
        const int64_t W = kTile;
        over_ranges(nel, W, [&](int64_t s_begin, int64_t s_end) {
        for (int64_t s = s_begin; s < s_end; s += W) {
            auto b = order.begin() + s, e = order.begin() + std::min(nel, s + W);
            std::stable_sort(b, e, [&](int32_t x, int32_t y) {
                if (owned(x) != owned(y)) return owned(x);                    // (the tile where the ghosts begin: owned nodes first)
                int dx = rows_of(x), dy = rows_of(y);
                if (dx != dy) return dx > dy;
                if (bnd_count[static_cast<size_t>(x)] != bnd_count[static_cast<size_t>(y)]) return bnd_count[static_cast<size_t>(x)] > bnd_count[static_cast<size_t>(y)];
                if (facing[static_cast<size_t>(x)] != facing[static_cast<size_t>(y)]) return facing[static_cast<size_t>(x)] < facing[static_cast<size_t>(y)];
                return cut_count[static_cast<size_t>(x)] > cut_count[static_cast<size_t>(y)];
            });
        }
        });
    }
    lap("degree / facing sort");
    P.old_of_new = order;
    P.new_of_old.assign(static_cast<size_t>(nel), 0);
    for (int64_t n = 0; n < nel; n++) P.new_of_old[static_cast<size_t>(order[static_cast<size_t>(n)])] = static_cast<int32_t>(n);

    // ---- slice geometry ----
    P.n_tiles = static_cast<int32_t>((nel + kTile - 1) / kTile);
    P.n_slices = P.n_tiles * (kTile / kSlice);          // the last tile may end in empty slices
    P.rows_int.assign(static_cast<size_t>(P.n_slices), 0);
    P.rows_bnd.assign(static_cast<size_t>(P.n_slices), 0);
    for (int64_t n = 0; n < nel; n++) {
        const int32_t old = order[static_cast<size_t>(n)];
        const int32_t s = static_cast<int32_t>(n / kSlice);
        const int32_t d = rows_of(old);
        P.rows_int[static_cast<size_t>(s)] = std::max(P.rows_int[static_cast<size_t>(s)], d);
        P.rows_bnd[static_cast<size_t>(s)] = std::max(P.rows_bnd[static_cast<size_t>(s)], bnd_count[static_cast<size_t>(old)]);
    }
    P.slice_row0.assign(static_cast<size_t>(P.n_slices) + 1, 0);
    for (int32_t s = 0; s < P.n_slices; s++)
        P.slice_row0[static_cast<size_t>(s) + 1] = P.slice_row0[static_cast<size_t>(s)] + P.rows_int[static_cast<size_t>(s)] + P.rows_bnd[static_cast<size_t>(s)];
    const int64_t rows = P.slice_row0.back();
    if (rows * kSlice >= (int64_t(1) << 31)) throw std::runtime_error("gather structure exceeds 2^31 entries");
    P.nbr.assign(static_cast<size_t>(rows) * kSlice, kCodePad);
    P.w.assign(static_cast<size_t>(rows) * kSlice, EdgeW{0.0, 0.0, 0.0, 0.0});

    // ---- fill, in original edge order = the reference's accumulation order ----
    std::vector<int32_t> fill_int(static_cast<size_t>(nel), 0), fill_bnd(static_cast<size_t>(nel), 0);
    std::vector<int32_t> entry_edge(static_cast<size_t>(rows) * kSlice, -1);   // internal entries: edge index - internal_start
    auto entry_index = [&](int32_t node_new, int32_t row_in_slice) {
        const int32_t s = node_new / kSlice, lane = node_new % kSlice;
        return (static_cast<int64_t>(P.slice_row0[static_cast<size_t>(s)]) + row_in_slice) * kSlice + lane;
    };
    // (every host thread walks ALL edges in order and fills the rows of the nodes in its own range of new ids: a node's entries
    //  keep the edge order whatever the number of threads, and a thread's writes stay in its own part of the arrays)
    std::atomic<int64_t> useful_total{0};
    over_ranges(nel, kTile, [&](int64_t n_begin, int64_t n_end) {
        int64_t useful_here = 0;
        for (int64_t e = L.internal_start; e < L.internal_start + L.n_internal; e++) {
            const mgcfd_edge &E = edges[static_cast<size_t>(e)];
            const int32_t a = P.new_of_old[static_cast<size_t>(E.a)], b = P.new_of_old[static_cast<size_t>(E.b)];
            const bool mine_a = a >= n_begin && a < n_end && owned(E.a), mine_b = b >= n_begin && b < n_end && owned(E.b);
            if (!mine_a && !mine_b) continue;
            const double ewt = std::sqrt(E.x * E.x + E.y * E.y + E.z * E.z);      // flux_kernel.elemfunc.c:27
            const double k = -ewt * kSmoothing * 0.5;                              // prefix of :130
            const double fx = -0.5 * E.x, fy = -0.5 * E.y, fz = -0.5 * E.z;       // :138-140
            if (mine_a) {
                int64_t ia = entry_index(a, fill_int[static_cast<size_t>(a)]++);
                P.nbr[static_cast<size_t>(ia)] = b;                                // this node is 'a'
                entry_edge[static_cast<size_t>(ia)] = static_cast<int32_t>(e - L.internal_start);
                P.w[static_cast<size_t>(ia)] = EdgeW{fx, fy, fz, k};
                useful_here++;
            }
            if (mine_b) {
                int64_t ib = entry_index(b, fill_int[static_cast<size_t>(b)]++);
                P.nbr[static_cast<size_t>(ib)] = a | kRoleB;                       // this node is 'b'
                entry_edge[static_cast<size_t>(ib)] = static_cast<int32_t>(e - L.internal_start);
                P.w[static_cast<size_t>(ib)] = EdgeW{-fx, -fy, -fz, k};            // x - f*y == x + (-f)*y exactly
                useful_here++;
            }
        }
        useful_total += useful_here;
    });
    int64_t useful = useful_total.load();
    auto add_face = [&](const mgcfd_edge &E, int32_t code, double scale) {
        if (!owned(E.b)) return;
        const int32_t b = P.new_of_old[static_cast<size_t>(E.b)];
        const int32_t s = b / kSlice;
        int64_t i = entry_index(b, P.rows_int[static_cast<size_t>(s)] + fill_bnd[static_cast<size_t>(b)]++);
        P.nbr[static_cast<size_t>(i)] = code;
        P.w[static_cast<size_t>(i)] = EdgeW{scale * E.x, scale * E.y, scale * E.z, 0.0};
    };
    for (int64_t e = L.boundary_start; e < L.boundary_start + L.n_boundary; e++) add_face(edges[static_cast<size_t>(e)], kCodeWall, 1.0);
    for (int64_t e = L.wall_start; e < L.wall_start + L.n_wall; e++) add_face(edges[static_cast<size_t>(e)], kCodeFar, 0.5);

    lap("rows filled");
    // ---- tiles: halo lists and 16-bit tile-local neighbour codes ----
    P.nbr16.assign(P.nbr.size(), static_cast<uint16_t>(kT16Pad));
    P.tile_halo_ptr.assign(static_cast<size_t>(P.n_tiles) + 1, 0);
    P.tile_ovf_ptr.assign(static_cast<size_t>(P.n_tiles) + 1, 0);
    P.tile_halo.clear();
    P.tile_ovf.clear();
    P.gat16.assign(P.nbr.size(), static_cast<uint16_t>(kT16Pad));
    P.te_chunk_ptr.assign(static_cast<size_t>(P.n_tiles) + 1, 0);
    P.te_count.assign(static_cast<size_t>(P.n_tiles), 0);
    P.te_slots.clear();
    P.te_w.clear();
    P.edge_once = true;
    {
        // The tiles are independent of each other: contiguous ranges of them are worked through by a thread each into a part of
        // its own (appended lists, prefix arrays counted from the range's start), and the parts are joined in tile order — the
        // plan is the same whatever the number of threads (round 4: this loop was most of the drop-in's start-up time).
        struct Part {
            std::vector<int32_t> tile_halo, tile_ovf, tile_halo_ptr, tile_ovf_ptr, te_chunk_ptr, hr_row0;
            std::vector<uint16_t> te_slots;
            std::vector<double> te_w, hr_w;
            std::vector<uint32_t> hr_code;
            bool edge_once = true, free_rows = true, free_wide = false;
            int32_t halo_max = 0, te_max = 0, hr_max_rows = 0, hr_max_tile_rows = 0;
            int64_t halo_total = 0, te_total = 0, halo_overflow_refs = 0, hr_entries = 0, hr_foreign = 0;
            std::exception_ptr error;
        };
        P.free_rows = true;
        P.hr_max_rows = P.hr_max_tile_rows = 0;
        P.free_wide = false;
        P.free_halo.assign(static_cast<size_t>(P.n_tiles) * kFreeHaloStride, -1);
        P.hr_row0.assign(static_cast<size_t>(P.n_slices) + 1, 0);
        P.hr_code.clear(); P.hr_w.clear(); P.hr_foreign = 0;
        P.hg16.assign(P.nbr.size(), static_cast<uint16_t>(kT16Pad));
        P.hr_entries = 0;
        const int32_t halo_cap = kTileCap - kTile;
        const int32_t ovf_cap = int32_t(kT16Far) - kTileCap;          // overflow slots a 15-bit code can name
        auto work = [&](int32_t t_begin, int32_t t_end, Part &Q) {
            try {
            const int32_t s_begin = t_begin * (kTile / kSlice);
            Q.tile_halo_ptr.assign(static_cast<size_t>(t_end - t_begin) + 1, 0);
            Q.tile_ovf_ptr.assign(static_cast<size_t>(t_end - t_begin) + 1, 0);
            Q.te_chunk_ptr.assign(static_cast<size_t>(t_end - t_begin) + 1, 0);
            Q.hr_row0.assign(static_cast<size_t>(t_end - t_begin) * (kTile / kSlice) + 1, 0);
            std::vector<int32_t> halo, tile_edges;
            std::vector<std::array<int64_t, 4>> half_ents;      // {edge, entry index, thread of the node, its half row or -1}
            std::vector<std::pair<int32_t, int32_t>> half_where;
            std::vector<int32_t> halo_all;                  // a tile's halo ids, ALL of them ascending (the order-free kernel's own numbering)
            for (int32_t t = t_begin; t < t_end; t++) {
                const int32_t base = t * kTile;
                const int32_t s0 = t * (kTile / kSlice), s1 = s0 + kTile / kSlice;
                const int64_t e0 = int64_t(P.slice_row0[static_cast<size_t>(s0)]) * kSlice;
                const int64_t e1 = int64_t(P.slice_row0[static_cast<size_t>(s1)]) * kSlice;
                halo.clear();
                for (int64_t e = e0; e < e1; e++) {
                    const int32_t code = P.nbr[static_cast<size_t>(e)];
                    if (code < 0) continue;
                    const int32_t id = code & kIdMask;
                    if (id < base || id >= base + kTile) halo.push_back(id);
                }
                std::sort(halo.begin(), halo.end());
                const int32_t n_refs = static_cast<int32_t>(halo.size());
                halo.erase(std::unique(halo.begin(), halo.end()), halo.end());
                const int32_t n_halo = static_cast<int32_t>(halo.size());
                const int32_t staged = std::min<int32_t>(n_halo, halo_cap);
                if (n_halo > staged) {
                    // more halo nodes than the LDS tile holds: stage the most referenced ones, leave the least
                    // referenced to the overflow table (each use of those is a gather from HBM); both parts ascending
                    std::vector<std::pair<int32_t, int32_t>> by_refs;                // (-references, id)
                    by_refs.reserve(static_cast<size_t>(n_halo));
                    {
                        std::vector<int32_t> all;
                        all.reserve(static_cast<size_t>(n_refs));
                        for (int64_t e = e0; e < e1; e++) {
                            const int32_t code = P.nbr[static_cast<size_t>(e)];
                            if (code < 0) continue;
                            const int32_t id = code & kIdMask;
                            if (id < base || id >= base + kTile) all.push_back(id);
                        }
                        std::sort(all.begin(), all.end());
                        for (size_t i = 0; i < all.size();) {
                            size_t j = i;
                            while (j < all.size() && all[j] == all[i]) j++;
                            by_refs.emplace_back(-static_cast<int32_t>(j - i), all[i]);
                            i = j;
                        }
                    }
                    std::sort(by_refs.begin(), by_refs.end());
                    for (int32_t k = 0; k < n_halo; k++) halo[static_cast<size_t>(k)] = by_refs[static_cast<size_t>(k)].second;
                    std::sort(halo.begin(), halo.begin() + staged);
                    std::sort(halo.begin() + staged, halo.end());
                }
                // position of a halo id in `halo` (staged part first, then the overflow part; each ascending)
                auto halo_pos = [&](int32_t id) -> int32_t {
                    auto it = std::lower_bound(halo.begin(), halo.begin() + staged, id);
                    if (it != halo.begin() + staged && *it == id) return static_cast<int32_t>(it - halo.begin());
                    return static_cast<int32_t>(std::lower_bound(halo.begin() + staged, halo.end(), id) - halo.begin());
                };
                if (n_halo - staged > ovf_cap) throw std::runtime_error("tile halo exceeds what 15-bit slots can address");
                Q.halo_total += n_halo;
                Q.halo_max = std::max<int32_t>(Q.halo_max, n_halo);
                for (int64_t e = e0; e < e1; e++) {
                    const int32_t code = P.nbr[static_cast<size_t>(e)];
                    uint32_t c16;
                    if (code == kCodeWall) c16 = kT16Wall;
                    else if (code == kCodeFar) c16 = kT16Far;
                    else if (code < 0) c16 = kT16Pad;
                    else {
                        const int32_t id = code & kIdMask;
                        uint32_t slot;
                        if (id >= base && id < base + kTile) slot = static_cast<uint32_t>(id - base);
                        else {
                            // the first `staged` halo ids live in LDS, the rest in the overflow table
                            const int32_t pos = halo_pos(id);
                            slot = static_cast<uint32_t>(kTile + pos);        // pos >= staged  =>  slot >= kTileCap
                            if (pos >= staged) Q.halo_overflow_refs++;
                        }
                        c16 = slot | ((code & kRoleB) ? kT16RoleB : 0u);
                    }
                    P.nbr16[static_cast<size_t>(e)] = static_cast<uint16_t>(c16);
                }
                // edge-once list: the tile's internal edges, once each, ascending original index.  A node's
                // incident edges keep their relative order in it, so summing a node's entries by position
                // is the reference's accumulation order.
                tile_edges.clear();
                for (int64_t e = e0; e < e1; e++)
                    if (entry_edge[static_cast<size_t>(e)] >= 0) tile_edges.push_back(entry_edge[static_cast<size_t>(e)]);
                std::sort(tile_edges.begin(), tile_edges.end());
                tile_edges.erase(std::unique(tile_edges.begin(), tile_edges.end()), tile_edges.end());
                const int32_t n_te = static_cast<int32_t>(tile_edges.size());
                P.te_count[static_cast<size_t>(t)] = n_te;
                Q.te_max = std::max(Q.te_max, n_te);
                Q.te_total += n_te;
                if (n_te > kMaxEdgeChunks * kEdgeChunk) Q.edge_once = false;
                if (Q.edge_once) {
                    auto slot_of = [&](int32_t id) -> uint16_t {
                        if (id >= base && id < base + kTile) return static_cast<uint16_t>(id - base);
                        return static_cast<uint16_t>(kTile + halo_pos(id));   // >= kTileCap: overflow table, as in nbr16
                    };
                    const size_t chunk0 = static_cast<size_t>(Q.te_chunk_ptr[static_cast<size_t>(t - t_begin)]);
                    const size_t n_chunks = (static_cast<size_t>(n_te) + kEdgeChunk - 1) / kEdgeChunk;
                    Q.te_slots.resize((chunk0 + n_chunks) * 2 * kEdgeChunk, static_cast<uint16_t>(kT16Pad));
                    Q.te_w.resize((chunk0 + n_chunks) * 4 * kEdgeChunk, 0.0);
                    for (int32_t p = 0; p < n_te; p++) {
                        const mgcfd_edge &E = edges[static_cast<size_t>(L.internal_start + tile_edges[static_cast<size_t>(p)])];
                        const size_t c = chunk0 + static_cast<size_t>(p / kEdgeChunk), ln = static_cast<size_t>(p % kEdgeChunk);
                        Q.te_slots[(c * 2 + 0) * kEdgeChunk + ln] = slot_of(P.new_of_old[static_cast<size_t>(E.a)]);
                        Q.te_slots[(c * 2 + 1) * kEdgeChunk + ln] = slot_of(P.new_of_old[static_cast<size_t>(E.b)]);
                        const double ewt = std::sqrt(E.x * E.x + E.y * E.y + E.z * E.z);
                        Q.te_w[(c * 4 + 0) * kEdgeChunk + ln] = -0.5 * E.x;
                        Q.te_w[(c * 4 + 1) * kEdgeChunk + ln] = -0.5 * E.y;
                        Q.te_w[(c * 4 + 2) * kEdgeChunk + ln] = -0.5 * E.z;
                        Q.te_w[(c * 4 + 3) * kEdgeChunk + ln] = -ewt * kSmoothing * 0.5;
                    }
                    for (int64_t e = e0; e < e1; e++) {
                        const int32_t ge = entry_edge[static_cast<size_t>(e)];
                        if (ge < 0) continue;
                        const uint32_t p = static_cast<uint32_t>(std::lower_bound(tile_edges.begin(), tile_edges.end(), ge) - tile_edges.begin());
                        P.gat16[static_cast<size_t>(e)] = static_cast<uint16_t>(p | ((P.nbr[static_cast<size_t>(e)] & kRoleB) ? kT16RoleB : 0u));
                    }
                    Q.te_chunk_ptr[static_cast<size_t>(t - t_begin) + 1] = static_cast<int32_t>(chunk0 + n_chunks);
                }
                // half rows: every entry of this tile whose node OWNS the evaluation of its edge (half_eval_a, decided before
                // the nodes were ordered) is placed in a (half row, lane) slot of the tile: in the node's own lane while its
                // slice has rows left (the evaluator then has its record in registers), otherwise in any lane with a free
                // slot ("foreign": that lane reads the owner's record from LDS as well).  A tile gets ceil(evaluations / 64)
                // half rows, spread over its slices in proportion to what their nodes own, so slots are ~98 % used.
                if (Q.free_rows) {
                    // the order-free kernel stages EVERY halo node (no overflow table): up to kHaloStride from the shared table, beyond
                    // that — up to kFreeHaloStride — from a table of its own, slots counted in the ascending list of all of them
                    halo_all.assign(halo.begin(), halo.end());
                    std::sort(halo_all.begin(), halo_all.end());
                    if (n_halo > kFreeHaloStride) Q.free_rows = false;
                    if (n_halo > staged) Q.free_wide = true;
                    if (Q.free_rows) std::copy(halo_all.begin(), halo_all.end(), P.free_halo.begin() + static_cast<size_t>(t) * kFreeHaloStride);
                    const int32_t n_here = static_cast<int32_t>(std::min<int64_t>(kTile, nel - base));
                    int32_t own[kTile] = {0};
                    half_ents.clear();
                    for (int32_t tid = 0; tid < n_here; tid++) {
                        const int32_t n = base + tid;
                        const int32_t s = n / kSlice, lane = n % kSlice;
                        for (int32_t r = 0; r < P.rows_int[static_cast<size_t>(s)]; r++) {
                            const int64_t e = (int64_t(P.slice_row0[static_cast<size_t>(s)]) + r) * kSlice + lane;
                            const int32_t ge = entry_edge[static_cast<size_t>(e)];
                            if (ge < 0) continue;
                            const int8_t who = half_eval_a[static_cast<size_t>(ge)];        // -1: both end points (each in its tile), 1: a, 0: b
                            const bool is_a = (P.nbr[static_cast<size_t>(e)] & kRoleB) == 0;
                            const bool evaluates = who < 0 || (who == 1) == is_a;
                            if (evaluates) own[tid]++;
                            half_ents.push_back({ge, e, tid, evaluates ? 1 : 0});
                        }
                    }
                    int32_t n_eval = 0, own_slice[kTile / kSlice] = {0}, lanes[kTile / kSlice] = {0};
                    for (int32_t tid = 0; tid < n_here; tid++) { n_eval += own[tid]; own_slice[tid / kSlice] += own[tid]; lanes[tid / kSlice]++; }
                    // rows per slice: proportional to what the slice's nodes own, then one more wherever the capacity is short
                    int32_t rows_h[kTile / kSlice] = {0};
                    // First with at most kHalfMaxRows rows per slice — what a lane of the ordered half-row kernel keeps in registers
                    // and what the order-free kernel requests up front: a slice that owns more spills its surplus into the other
                    // slices' lanes (foreign entries) and the level stays eligible for k_flux_half.  Only a tile whose evaluations do
                    // not fit 5 rows x its lanes gets longer slices (up to kFreeMaxRows: the order-free kernel walks them in a loop).
                    // (Round 3 went straight to the long slices and so took non-uniform levels away from k_flux_half: advisor finding.)
                    auto distribute = [&](int32_t row_cap) {
                        int32_t cap_total = 0;
                        for (int32_t sl = 0; sl < kTile / kSlice; sl++) {
                            rows_h[sl] = lanes[sl] ? std::min<int32_t>(row_cap, own_slice[sl] / lanes[sl]) : 0;    // floor of the slice's mean
                            cap_total += rows_h[sl] * lanes[sl];
                        }
                        while (cap_total < n_eval) {
                            int32_t best = -1; double need = -1e300;
                            for (int32_t sl = 0; sl < kTile / kSlice; sl++) {
                                if (!lanes[sl] || rows_h[sl] >= row_cap) continue;
                                const double d = double(own_slice[sl]) / lanes[sl] - rows_h[sl];      // how far the slice's mean is above its rows
                                if (d > need) { need = d; best = sl; }
                            }
                            if (best < 0) return false;
                            rows_h[best]++; cap_total += lanes[best];
                        }
                        return true;
                    };
                    if (!distribute(kHalfMaxRows) && !distribute(kFreeMaxRows)) Q.free_rows = false;
                    for (int32_t sl = 0; sl < kTile / kSlice; sl++)
                        Q.hr_row0[static_cast<size_t>(s0 - s_begin + sl) + 1] = Q.hr_row0[static_cast<size_t>(s0 - s_begin + sl)] + rows_h[sl];
                    for (int32_t sl = 0; sl < kTile / kSlice; sl++) Q.hr_max_rows = std::max(Q.hr_max_rows, rows_h[sl]);
                    Q.hr_max_tile_rows = std::max(Q.hr_max_tile_rows, Q.hr_row0[static_cast<size_t>(s1 - s_begin)] - Q.hr_row0[static_cast<size_t>(s0 - s_begin)]);
                    if (Q.free_rows) {
                        const size_t rows_end = static_cast<size_t>(Q.hr_row0[static_cast<size_t>(s1 - s_begin)]);
                        Q.hr_code.resize(rows_end * kSlice, kHalfPad);
                        Q.hr_w.resize(rows_end * 3 * kSlice, 0.0);
                        int32_t used[kTile] = {0};
                        half_where.clear();                                   // (edge, position of its flux terms) of this tile's evaluations
                        auto place = [&](const std::array<int64_t, 4> &en, int32_t host) {
                            const int32_t owner = static_cast<int32_t>(en[2]);
                            const int32_t sl = host / kSlice, lane = host % kSlice, j = used[host]++;
                            const size_t hrow = static_cast<size_t>(Q.hr_row0[static_cast<size_t>(s0 - s_begin + sl)]) + static_cast<size_t>(j);
                            const int64_t e = en[1];
                            const bool only = half_eval_a[static_cast<size_t>(en[0])] >= 0;     // (an edge inside the tile: nobody else evaluates it)
                            // (the other end's LDS slot: own nodes as in nbr16; a halo node by its position among ALL the tile's halo ids —
                            //  the same slot as nbr16's wherever the tile has no overflow entries)
                            const int32_t other = P.nbr[static_cast<size_t>(e)] & kIdMask;
                            const uint32_t oslot = (other >= base && other < base + kTile) ? uint32_t(other - base)
                                                 : uint32_t(kTile + (std::lower_bound(halo_all.begin(), halo_all.end(), other) - halo_all.begin()));
                            const uint32_t c16 = oslot | ((P.nbr[static_cast<size_t>(e)] & kRoleB) ? kT16RoleB : 0u);
                            Q.hr_code[hrow * kSlice + lane] = c16 | (uint32_t(owner) << 16) | (host != owner ? kHalfForeign : 0u)
                                                              | (only ? kHalfMirror : 0u);
                            const EdgeW &W = P.w[static_cast<size_t>(e)];
                            Q.hr_w[(hrow * 3 + 0) * kSlice + lane] = W.x;
                            Q.hr_w[(hrow * 3 + 1) * kSlice + lane] = W.y;
                            Q.hr_w[(hrow * 3 + 2) * kSlice + lane] = W.z;
                            const int32_t pos = static_cast<int32_t>(hrow - static_cast<size_t>(Q.hr_row0[static_cast<size_t>(s0 - s_begin)])) * kSlice + lane;
                            P.hg16[static_cast<size_t>(e)] = static_cast<uint16_t>(pos);
                            half_where.emplace_back(static_cast<int32_t>(en[0]), pos);
                            Q.hr_entries++;
                            if (host != owner) Q.hr_foreign++;
                        };
                        // own lane first (entries come node by node, in the node's row order) ...
                        for (auto &en : half_ents) {
                            if (en[3] != 1) continue;
                            const int32_t tid = static_cast<int32_t>(en[2]);
                            if (used[tid] < rows_h[tid / kSlice]) { place(en, tid); en[3] = 2; }
                        }
                        // ... the rest wherever a slot is free
                        int32_t host = 0;
                        for (auto &en : half_ents) {
                            if (en[3] != 1) continue;
                            while (host < n_here && used[host] >= rows_h[host / kSlice]) host++;
                            if (host >= n_here) throw std::logic_error("half rows: no free slot left in the tile");
                            place(en, host);
                            en[3] = 2;
                        }
                        std::sort(half_where.begin(), half_where.end());
                        for (const auto &en : half_ents) {
                            if (en[3] != 0) continue;                        // the other end point owns the evaluation: its position, negated
                            auto it = std::lower_bound(half_where.begin(), half_where.end(), std::make_pair(static_cast<int32_t>(en[0]), int32_t(-1)));
                            if (it == half_where.end() || it->first != static_cast<int32_t>(en[0])) throw std::logic_error("half rows: an edge without an evaluator in its tile");
                            P.hg16[static_cast<size_t>(en[1])] = static_cast<uint16_t>(uint32_t(it->second) | kT16RoleB);
                        }
                    }
                }
                Q.tile_halo.insert(Q.tile_halo.end(), halo.begin(), halo.begin() + staged);
                Q.tile_halo_ptr[static_cast<size_t>(t - t_begin) + 1] = static_cast<int32_t>(Q.tile_halo.size());
                Q.tile_ovf.insert(Q.tile_ovf.end(), halo.begin() + staged, halo.end());
                Q.tile_ovf_ptr[static_cast<size_t>(t - t_begin) + 1] = static_cast<int32_t>(Q.tile_ovf.size());
            }
            } catch (...) { Q.error = std::current_exception(); }
        };
        int n_threads = opt.threads > 0 ? opt.threads : int(std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency())));
        n_threads = std::max(1, std::min<int>(n_threads, P.n_tiles / 64));          // (a range of fewer than 64 tiles is not worth a thread)
        if (const char *o = std::getenv("MGCFD_PLAN_THREADS")) n_threads = std::max(1, std::min<int>(std::atoi(o), P.n_tiles));   // (tests: any split)
        std::vector<Part> parts(static_cast<size_t>(n_threads));
        {
            std::vector<std::thread> workers;
            auto range_begin = [&](int k) { return static_cast<int32_t>(int64_t(P.n_tiles) * k / n_threads); };
            for (int k = 1; k < n_threads; k++) workers.emplace_back(work, range_begin(k), range_begin(k + 1), std::ref(parts[static_cast<size_t>(k)]));
            work(range_begin(0), range_begin(1), parts[0]);
            for (std::thread &w : workers) w.join();
        }
        for (const Part &Q : parts) if (Q.error) std::rethrow_exception(Q.error);     // (the lowest tile range's error first)
        // join the parts in tile order
        int64_t halo_total = 0, te_total = 0;
        {
            int32_t t0 = 0;
            for (int k = 0; k < n_threads; k++) {
                const Part &Q = parts[static_cast<size_t>(k)];
                const int32_t nt = static_cast<int32_t>(Q.tile_halo_ptr.size()) - 1;
                const int32_t halo0 = static_cast<int32_t>(P.tile_halo.size()), ovf0 = static_cast<int32_t>(P.tile_ovf.size());
                const int32_t chunk0 = P.te_chunk_ptr[static_cast<size_t>(t0)], row0 = P.hr_row0[static_cast<size_t>(t0) * (kTile / kSlice)];
                for (int32_t t = 0; t < nt; t++) {
                    P.tile_halo_ptr[static_cast<size_t>(t0 + t) + 1] = halo0 + Q.tile_halo_ptr[static_cast<size_t>(t) + 1];
                    P.tile_ovf_ptr[static_cast<size_t>(t0 + t) + 1] = ovf0 + Q.tile_ovf_ptr[static_cast<size_t>(t) + 1];
                    P.te_chunk_ptr[static_cast<size_t>(t0 + t) + 1] = chunk0 + Q.te_chunk_ptr[static_cast<size_t>(t) + 1];
                }
                for (size_t q = 1; q < Q.hr_row0.size(); q++) P.hr_row0[static_cast<size_t>(t0) * (kTile / kSlice) + q] = row0 + Q.hr_row0[q];
                P.tile_halo.insert(P.tile_halo.end(), Q.tile_halo.begin(), Q.tile_halo.end());
                P.tile_ovf.insert(P.tile_ovf.end(), Q.tile_ovf.begin(), Q.tile_ovf.end());
                // (a part's lists end where its last tile's chunks / half rows end: resize() inside the loop keeps them exact)
                P.te_slots.insert(P.te_slots.end(), Q.te_slots.begin(), Q.te_slots.end());
                P.te_w.insert(P.te_w.end(), Q.te_w.begin(), Q.te_w.end());
                P.hr_code.insert(P.hr_code.end(), Q.hr_code.begin(), Q.hr_code.end());
                P.hr_w.insert(P.hr_w.end(), Q.hr_w.begin(), Q.hr_w.end());
                P.edge_once = P.edge_once && Q.edge_once;
                if (P.free_rows) {
                    // (one thread stops looking at half rows at the first tile that cannot have them; a part stopped at ITS first
                    //  such tile, so the parts up to and including the first failing one are what one thread would have seen)
                    P.free_wide = P.free_wide || Q.free_wide;
                    P.hr_max_rows = std::max(P.hr_max_rows, Q.hr_max_rows);
                    P.hr_max_tile_rows = std::max(P.hr_max_tile_rows, Q.hr_max_tile_rows);
                    P.hr_entries += Q.hr_entries;
                    P.hr_foreign += Q.hr_foreign;
                    P.free_rows = Q.free_rows;
                }
                P.halo_max = std::max(P.halo_max, Q.halo_max);
                P.te_max = std::max(P.te_max, Q.te_max);
                P.halo_overflow_refs += Q.halo_overflow_refs;
                halo_total += Q.halo_total;
                te_total += Q.te_total;
                t0 += nt;
            }
        }
        P.halo_mean = P.n_tiles ? double(halo_total) / double(P.n_tiles) : 0.0;
        P.halo_total = halo_total;
        P.te_mean = P.n_tiles ? double(te_total) / double(P.n_tiles) : 0.0;
        if (!P.edge_once) {
            P.te_slots.clear(); P.te_w.clear(); P.gat16.clear();
            P.te_chunk_ptr.assign(static_cast<size_t>(P.n_tiles) + 1, 0);
        }
    }

    lap("tile loop");
    // two-phase design point: per-edge arrays in original order and the rows' edge references
    {
        const size_t ne = static_cast<size_t>(L.n_internal);
        P.fe_ab.assign(2 * ne, 0);
        P.fe_w.assign(4 * ne, 0.0);
        for (size_t k = 0; k < ne; k++) {
            const mgcfd_edge &E = edges[static_cast<size_t>(L.internal_start) + k];
            P.fe_ab[k] = P.new_of_old[static_cast<size_t>(E.a)];
            P.fe_ab[ne + k] = P.new_of_old[static_cast<size_t>(E.b)];
            const double ewt = std::sqrt(E.x * E.x + E.y * E.y + E.z * E.z);
            P.fe_w[k] = -0.5 * E.x; P.fe_w[ne + k] = -0.5 * E.y; P.fe_w[2 * ne + k] = -0.5 * E.z;
            P.fe_w[3 * ne + k] = -ewt * kSmoothing * 0.5;
        }
        P.row_edge.assign(P.nbr.size(), -1);
        for (size_t e = 0; e < entry_edge.size(); e++)
            if (entry_edge[e] >= 0)
                P.row_edge[e] = static_cast<int32_t>(static_cast<uint32_t>(entry_edge[e]) | ((P.nbr[e] & kRoleB) ? 0x80000000u : 0u));
    }
    lap("fission arrays");
    // ---- long rows: per tile, the row limit that minimises the estimated time of its workgroup ----
    // Units: one pair of rows of the per-node loop = 1.  The four waves walk their slices side by side, so the loop
    // costs the longest slice; the workgroup's list costs a fixed part (barrier, scratch round trip), a part per round
    // of 256 entries and a part per entry of the longest per-node run of ordered adds.
    P.rows_main = P.rows_int;
    P.tail_tile_ptr.assign(static_cast<size_t>(P.n_tiles) + 1, 0);
    P.tail_begin.assign(static_cast<size_t>(nel), 0);
    P.tail_count.assign(static_cast<size_t>(nel), 0);
    P.tail_rec.clear();
    P.has_tail = false;
    auto push_tail = [&](const EdgeW &w, uint32_t owner, uint32_t code) {
        const uint64_t word = uint64_t(owner) | (uint64_t(code) << 16);
        double as_double;
        std::memcpy(&as_double, &word, sizeof(as_double));
        const double rec[6] = {w.x, w.y, w.z, w.k, as_double, 0.0};
        P.tail_rec.insert(P.tail_rec.end(), rec, rec + 6);
    };
    // fitted on a tetrahedral level (tools/tet_mesh_bench.py; the optimum is flat): a round of the list costs more than
    // a row pair of the loop (single entries, the owner's record and flux terms fetched again, the hand-over)
    const double c_fixed = 1.5, c_round = 1.5, c_add = 0.1;
    if (opt.long_rows) {
        for (int32_t t = 0; t < P.n_tiles; t++) {
            const int32_t base = t * kTile, s0 = t * (kTile / kSlice);
            const int32_t n_here = static_cast<int32_t>(std::min<int64_t>(kTile, nel - base));
            int32_t deg[kTile], max_deg = 0;
            for (int32_t k = 0; k < n_here; k++) { deg[k] = rows_of(order[static_cast<size_t>(base + k)]); max_deg = std::max(max_deg, deg[k]); }
            auto cost = [&](int32_t T) {
                int64_t tail = 0;
                for (int32_t k = 0; k < n_here; k++) tail += std::max(0, deg[k] - T);
                int32_t longest = 0;
                for (int32_t q = 0; q < kTile / kSlice; q++) longest = std::max(longest, std::min(T, P.rows_int[static_cast<size_t>(s0 + q)]));
                return double((longest + 1) / 2) +
                       (tail > 0 ? c_fixed + c_round * double((tail + kTile - 1) / kTile) + c_add * double(max_deg - T) : 0.0);
            };
            int32_t best_T = max_deg;
            double best = cost(max_deg);
            for (int32_t T = 2; T < max_deg; T += 2) {
                const double c = cost(T);
                if (c < best - 0.5) { best = c; best_T = T; }          // (only for a clear gain)
            }
            P.tail_tile_ptr[static_cast<size_t>(t)] = static_cast<int32_t>(P.tail_rec.size() / 6);
            // A node's own cut: the tile's row limit, or the first entry whose neighbour did not fit the LDS tile —
            // in the per-node loop such an entry is a gather from HBM that the whole wave waits for; in the
            // workgroup's list it is one of many loads in flight.  Everything from the cut on goes to the list
            // (the order of the node's sum is kept: loop rows first, then its list entries in row order).
            int32_t cut[kTile];
            bool any = false;
            for (int32_t k = 0; k < n_here; k++) {
                cut[k] = std::min(best_T, deg[k]);
                for (int32_t r = 0; r < cut[k]; r++)
                    if ((P.nbr16[static_cast<size_t>(entry_index(base + k, r))] & kT16SlotMask) >= uint32_t(kTileCap)) { cut[k] = r; break; }
                any = any || cut[k] < deg[k];
            }
            if (any) {
                P.has_tail = true;
                for (int32_t q = 0; q < kTile / kSlice; q++) P.rows_main[static_cast<size_t>(s0 + q)] = 0;
                for (int32_t k = 0; k < n_here; k++) {
                    int32_t &rm = P.rows_main[static_cast<size_t>(s0 + k / kSlice)];
                    rm = std::max(rm, cut[k]);
                    P.tail_begin[static_cast<size_t>(base + k)] = static_cast<int32_t>(P.tail_rec.size() / 6);
                    P.tail_count[static_cast<size_t>(base + k)] = deg[k] - cut[k];
                    for (int32_t r = cut[k]; r < deg[k]; r++) {
                        const int64_t e = entry_index(base + k, r);
                        push_tail(P.w[static_cast<size_t>(e)], static_cast<uint32_t>(k), P.nbr16[static_cast<size_t>(e)]);
                        P.nbr16[static_cast<size_t>(e)] = static_cast<uint16_t>(kT16Pad);      // (the weights stay: indirect_rw reads them)
                    }
                }
                while ((P.tail_rec.size() / 6) % 8)                     // whole 128-byte lines of the scratch per tile (8 x 48 B = 3 lines)
                    push_tail(EdgeW{0.0, 0.0, 0.0, 0.0}, kT16Pad, kT16Pad);
            }
        }
    }
    P.tail_tile_ptr[static_cast<size_t>(P.n_tiles)] = static_cast<int32_t>(P.tail_rec.size() / 6);
    P.tail_total = static_cast<int64_t>(P.tail_rec.size() / 6);
    if (!P.has_tail) { P.tail_begin.clear(); P.tail_count.clear(); }
    // the ordered half-row kernel: levels without long rows whose every slice fits the per-thread row limit and every tile the
    // flux terms' LDS image; the order-free kernel (free_rows) takes any row count
    P.half = P.free_rows && !P.free_wide && !P.has_tail && P.hr_max_rows <= kHalfMaxRows && P.hr_max_tile_rows <= kHalfTileRows;
    if (!P.free_rows || !P.free_wide) { P.free_halo.clear(); P.free_halo.shrink_to_fit(); }
    if (!P.free_rows) { P.hr_row0.clear(); P.hr_code.clear(); P.hr_w.clear(); P.hg16.clear(); P.hr_entries = 0; P.hr_foreign = 0; }
    else P.hr_padding = int64_t(P.hr_row0.back()) * kSlice - P.hr_entries;

    lap("long rows");
    P.n_internal_entries = useful;
    int64_t int_slots = 0;
    for (int32_t s = 0; s < P.n_slices; s++) int_slots += static_cast<int64_t>(P.rows_int[static_cast<size_t>(s)]) * kSlice;
    P.pad_fraction = useful > 0 ? double(int_slots - useful) / double(useful) : 0.0;
    P.pad_entries = int_slots - useful;
}

void build_transfer_plan(const mgcfd_level_desc &F, const std::vector<mgcfd_edge> &fine_edges,
                         const double *coarse_coords, int64_t nel_coarse,
                         const std::vector<int32_t> &coarse_new_of_old, LevelPlan &P,
                         const int64_t *child_order_key, int64_t n_owned_fine)
{
    if (!F.mg_map) throw std::runtime_error("level has no multigrid map");
    const int64_t nel = F.nel, mgc = F.mgc;
    if (mgc > nel) throw std::runtime_error("multigrid map longer than the fine level");
    for (int64_t i = 0; i < mgc; i++)
        if (F.mg_map[i] < 0 || F.mg_map[i] >= nel_coarse) throw std::runtime_error("multigrid map entry out of range");

    // ---- restriction: children of every coarse node, ascending original fine id ----
    P.child_ptr.assign(static_cast<size_t>(nel_coarse) + 1, 0);
    for (int64_t i = 0; i < mgc; i++) P.child_ptr[static_cast<size_t>(coarse_new_of_old[static_cast<size_t>(F.mg_map[i])]) + 1]++;
    std::partial_sum(P.child_ptr.begin(), P.child_ptr.end(), P.child_ptr.begin());
    P.child.assign(static_cast<size_t>(mgc), 0);
    {
        std::vector<int32_t> fill(P.child_ptr.begin(), P.child_ptr.end() - 1);
        std::vector<int64_t> order(static_cast<size_t>(mgc));
        std::iota(order.begin(), order.end(), int64_t(0));
        if (child_order_key)
            std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return child_order_key[x] < child_order_key[y]; });
        for (int64_t k = 0; k < mgc; k++) {
            const int64_t i = order[static_cast<size_t>(k)];
            const int32_t c = coarse_new_of_old[static_cast<size_t>(F.mg_map[i])];
            P.child[static_cast<size_t>(fill[static_cast<size_t>(c)]++)] = P.new_of_old[static_cast<size_t>(i)];
        }
    }
    const int64_t n_owned = (n_owned_fine < 0 || n_owned_fine > nel) ? nel : n_owned_fine;

    // ---- prolongation weights (mg_loops.cpp:730-812), static because geometry is ----
    if (mgc < nel) throw std::runtime_error("prolongation needs a parent for every fine node (mgc < nel)");
    if (!F.coords || !coarse_coords) throw std::runtime_error("prolongation needs node coordinates on both levels");
    P.pro.assign(P.nbr.size(), ProlongW{0.0, 0.0, 0, 0});
    P.pro_parent.assign(static_cast<size_t>(nel), 0);
    P.pro_wsum.assign(static_cast<size_t>(nel), 0.0);
    std::vector<char> is_coincident(static_cast<size_t>(nel), 0);
    for (int64_t i = 0; i < nel; i++) {
        const int64_t p_old = F.mg_map[i];
        const int32_t p_new = coarse_new_of_old[static_cast<size_t>(p_old)];
        const bool same = coincident(F.coords + 3 * i, coarse_coords + 3 * p_old);
        is_coincident[static_cast<size_t>(i)] = same;
        const int32_t n = P.new_of_old[static_cast<size_t>(i)];
        P.pro_parent[static_cast<size_t>(n)] = same ? ~p_new : p_new;
        if (same) P.pro_wsum[static_cast<size_t>(n)] = 1.0;
    }
    std::vector<int32_t> fill(static_cast<size_t>(nel), 0);
    auto entry_index = [&](int32_t node_new, int32_t row_in_slice) {
        const int32_t s = node_new / kSlice, lane = node_new % kSlice;
        return (static_cast<int64_t>(P.slice_row0[static_cast<size_t>(s)]) + row_in_slice) * kSlice + lane;
    };
    // The four inverse distances of every edge (a square root and a division each: most of this function's time) are
    // computed by several host threads over ranges of edges; the pass that places them — a node's sum of weights grows in
    // edge order, as the reference's does — stays serial.
    const int64_t n_int = F.n_internal;
    std::vector<double> invd(static_cast<size_t>(n_int) * 4, 0.0);          // per edge: a own, a other, b own, b other
    {
        int n_threads = int(std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency())));
        n_threads = std::max(1, std::min<int>(n_threads, int(n_int / 50000)));
        if (const char *o = std::getenv("MGCFD_PLAN_THREADS")) n_threads = std::max(1, std::atoi(o));
        auto work = [&](int64_t k0, int64_t k1) {
            for (int64_t k = k0; k < k1; k++) {
                const mgcfd_edge &E = fine_edges[static_cast<size_t>(F.internal_start + k)];
                const int64_t a2 = E.a, b2 = E.b;
                const int64_t a1 = F.mg_map[a2], b1 = F.mg_map[b2];
                const double *ca1 = coarse_coords + 3 * a1, *cb1 = coarse_coords + 3 * b1;
                const double *ca2 = F.coords + 3 * a2, *cb2 = F.coords + 3 * b2;
                double *w = invd.data() + 4 * k;
                if (a2 < n_owned && !is_coincident[static_cast<size_t>(a2)]) { w[0] = inv_distance(ca2, ca1); w[1] = inv_distance(cb1, ca2); }   // :754, :768
                if (b2 < n_owned && !is_coincident[static_cast<size_t>(b2)]) { w[2] = inv_distance(cb2, cb1); w[3] = inv_distance(ca1, cb2); }   // :792, :806
            }
        };
        std::vector<std::thread> workers;
        for (int t = 1; t < n_threads; t++) workers.emplace_back(work, n_int * t / n_threads, n_int * (t + 1) / n_threads);
        work(0, n_int / n_threads);
        for (std::thread &w : workers) w.join();
    }
    for (int64_t e = F.internal_start; e < F.internal_start + F.n_internal; e++) {
        const int64_t a2 = fine_edges[static_cast<size_t>(e)].a, b2 = fine_edges[static_cast<size_t>(e)].b;
        const int64_t a1 = F.mg_map[a2], b1 = F.mg_map[b2];
        const int32_t a1n = coarse_new_of_old[static_cast<size_t>(a1)], b1n = coarse_new_of_old[static_cast<size_t>(b1)];
        const int32_t an = P.new_of_old[static_cast<size_t>(a2)], bn = P.new_of_old[static_cast<size_t>(b2)];
        const double *iw = invd.data() + 4 * (e - F.internal_start);
        // a2's entry: own parent a1, then b1 (ghosts of a partitioned level hold no rows)
        if (a2 < n_owned) {
            ProlongW w{0.0, 0.0, a1n, b1n};
            if (!is_coincident[static_cast<size_t>(a2)]) {
                w.w_own = iw[0];                       // :754
                w.w_other = iw[1];                     // :768
                P.pro_wsum[static_cast<size_t>(an)] += w.w_own;
                P.pro_wsum[static_cast<size_t>(an)] += w.w_other;
            }
            P.pro[static_cast<size_t>(entry_index(an, fill[static_cast<size_t>(an)]++))] = w;
        }
        // b2's entry: own parent b1, then "a1" — which the reference reads from b1 (:805-809)
        if (b2 < n_owned) {
            ProlongW w{0.0, 0.0, b1n, b1n};
            if (!is_coincident[static_cast<size_t>(b2)]) {
                w.w_own = iw[2];                       // :792
                w.w_other = iw[3];                     // :806
                P.pro_wsum[static_cast<size_t>(bn)] += w.w_own;
                P.pro_wsum[static_cast<size_t>(bn)] += w.w_other;
            }
            P.pro[static_cast<size_t>(entry_index(bn, fill[static_cast<size_t>(bn)]++))] = w;
        }
    }
    // A coincident node that no internal edge touches never gets its w_sums/res2_wavg assigned
    // in the reference (both stay 0 => 0/0); express it as a plain node with no entries.
    for (int64_t n = 0; n < nel; n++)
        if (P.pro_parent[static_cast<size_t>(n)] < 0 && fill[static_cast<size_t>(n)] == 0) {
            P.pro_parent[static_cast<size_t>(n)] = ~P.pro_parent[static_cast<size_t>(n)];
            P.pro_wsum[static_cast<size_t>(n)] = 0.0;
        }

    // ---- tiled form: per fine tile, the distinct coarse nodes its nodes and entries refer to ----
    P.pro_tiled = true;
    P.pro_tile_n.assign(static_cast<size_t>(P.n_tiles), 0);
    P.pro_tile_ids.assign(static_cast<size_t>(P.n_tiles) * kProCap, -1);
    P.pro_s16.assign(P.pro.size(), 0);
    P.pro_own16.assign(static_cast<size_t>(nel), 0);
    auto parent_of = [&](int64_t n) { const int32_t p = P.pro_parent[static_cast<size_t>(n)]; return p < 0 ? ~p : p; };
    // (the tiles write disjoint parts of every array: ranges of them on host threads; a tile that refers to more coarse nodes
    //  than the LDS image holds switches the tiled form off for the level, whoever finds it)
    std::atomic<bool> tiled_ok{true};
    auto tile_work = [&](int32_t t_begin, int32_t t_end) {
    std::vector<int32_t> ids;
    for (int32_t t = t_begin; t < t_end && tiled_ok.load(std::memory_order_relaxed); t++) {
        const int64_t n0 = int64_t(t) * kTile, n1 = std::min<int64_t>(nel, n0 + kTile);
        const int32_t s0 = t * (kTile / kSlice), s1 = s0 + kTile / kSlice;
        ids.clear();
        for (int64_t n = n0; n < n1; n++) ids.push_back(parent_of(n));
        for (int32_t sl = s0; sl < s1; sl++) {
            const int64_t e0 = int64_t(P.slice_row0[static_cast<size_t>(sl)]) * kSlice;
            const int64_t e1 = e0 + int64_t(P.rows_int[static_cast<size_t>(sl)]) * kSlice;
            for (int64_t e = e0; e < e1; e++) {
                const ProlongW &w = P.pro[static_cast<size_t>(e)];
                if (w.w_own != 0.0 || w.w_other != 0.0) ids.push_back(w.p_other);
            }
        }
        std::sort(ids.begin(), ids.end());
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        if (ids.size() > static_cast<size_t>(kProCap)) { tiled_ok.store(false, std::memory_order_relaxed); break; }
        P.pro_tile_n[static_cast<size_t>(t)] = static_cast<int32_t>(ids.size());
        std::copy(ids.begin(), ids.end(), P.pro_tile_ids.begin() + static_cast<size_t>(t) * kProCap);
        auto pos = [&](int32_t id) { return static_cast<uint16_t>(std::lower_bound(ids.begin(), ids.end(), id) - ids.begin()); };
        for (int64_t n = n0; n < n1; n++) P.pro_own16[static_cast<size_t>(n)] = pos(parent_of(n));
        for (int32_t sl = s0; sl < s1; sl++) {
            const int64_t e0 = int64_t(P.slice_row0[static_cast<size_t>(sl)]) * kSlice;
            const int64_t e1 = e0 + int64_t(P.rows_int[static_cast<size_t>(sl)]) * kSlice;
            for (int64_t e = e0; e < e1; e++) {
                const ProlongW &w = P.pro[static_cast<size_t>(e)];
                if (w.w_own != 0.0 || w.w_other != 0.0) P.pro_s16[static_cast<size_t>(e)] = pos(w.p_other);
            }
        }
    }
    };
    {
        int n_threads = int(std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency())));
        n_threads = std::max(1, std::min<int>(n_threads, P.n_tiles / 64));
        if (const char *o = std::getenv("MGCFD_PLAN_THREADS")) n_threads = std::max(1, std::min<int>(std::atoi(o), P.n_tiles));
        std::vector<std::thread> workers;
        auto range_begin = [&](int k) { return static_cast<int32_t>(int64_t(P.n_tiles) * k / n_threads); };
        for (int k = 1; k < n_threads; k++) workers.emplace_back(tile_work, range_begin(k), range_begin(k + 1));
        tile_work(range_begin(0), range_begin(1));
        for (std::thread &w : workers) w.join();
    }
    P.pro_tiled = tiled_ok.load();
    if (!P.pro_tiled) { P.pro_tile_n.clear(); P.pro_tile_ids.clear(); P.pro_s16.clear(); P.pro_own16.clear(); }
}

// ------------------------------------------------------------------------------------------
// Bounds audit of a freshly built plan (before the solver drops the host copies): every index the kernels form from
// the plan's tables — LDS slots, halo and overflow positions, half-row owners, list entries, children, staged coarse
// nodes — against the size of what it indexes, as kernels.hip forms it (k_flux_tile and its role-5 form, k_flux_half,
// k_flux_free, k_flux_edge_once, k_restrict, k_prolong_tile).  Returns "" when everything is in range, else one line
// per kind of violation (the first few of each).  Host only: tests/test_host_plan_audit.py, tools/plan_stats.cpp.
// ------------------------------------------------------------------------------------------
// A digest of every array the per-tile part of the plan produces (FNV-1a over their bytes): the plan must not depend on how many
// host threads built it (tests/test_host_plan_audit.py; mgcfd_plan_audit appends it to its report when MGCFD_PLAN_DIGEST is set).
uint64_t plan_digest(const LevelPlan &P)
{
    uint64_t h = 1469598103934665603ull;
    auto eat = [&](const void *p, size_t n) {
        const unsigned char *b = static_cast<const unsigned char *>(p);
        for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    };
    auto vec = [&](const auto &v) { const uint64_t n = v.size(); eat(&n, sizeof(n)); if (n) eat(v.data(), n * sizeof(v[0])); };
    vec(P.old_of_new); vec(P.slice_row0); vec(P.rows_int); vec(P.rows_bnd); vec(P.nbr); vec(P.nbr16);
    vec(P.tile_halo_ptr); vec(P.tile_halo); vec(P.tile_ovf_ptr); vec(P.tile_ovf);
    vec(P.te_chunk_ptr); vec(P.te_count); vec(P.te_slots); vec(P.te_w); vec(P.gat16);
    vec(P.free_halo); vec(P.hr_row0); vec(P.hr_code); vec(P.hr_w); vec(P.hg16);
    vec(P.rows_main); vec(P.tail_tile_ptr); vec(P.tail_rec); vec(P.tail_begin); vec(P.tail_count);
    // ... and the transfer plan (build_transfer_plan: inverse distances and its per-tile part on host threads as well)
    vec(P.child_ptr); vec(P.child); vec(P.pro); vec(P.pro_parent); vec(P.pro_wsum); vec(P.pro_tile_n); vec(P.pro_tile_ids); vec(P.pro_s16); vec(P.pro_own16);
    const int64_t scalars[] = {P.halo_max, P.te_max, P.hr_max_rows, P.hr_max_tile_rows, P.halo_overflow_refs, P.hr_entries, P.hr_foreign, P.halo_total,
                               int64_t(P.edge_once), int64_t(P.free_rows), int64_t(P.free_wide), int64_t(P.half), int64_t(P.has_tail)};
    eat(scalars, sizeof(scalars));
    return h;
}

std::string audit_level_plan(const mgcfd_level_desc &L, const LevelPlan &P, int64_t nel_coarse)
{
    std::string rep;
    int shown[32] = {0};
    auto bad = [&](int kind, const std::string &what) { if (shown[kind]++ < 3) rep += what + "\n"; };
    const int64_t nel = L.nel;
    auto S = [](int64_t v) { return std::to_string(v); };
    if (P.nel != nel) bad(0, "plan.nel " + S(P.nel) + " != level nel " + S(nel));
    if (int64_t(P.old_of_new.size()) != nel || int64_t(P.new_of_old.size()) != nel) { bad(0, "permutation arrays are not nel long"); return rep; }
    for (int64_t n = 0; n < nel; n++) {
        const int32_t o = P.old_of_new[size_t(n)];
        if (o < 0 || o >= nel || P.new_of_old[size_t(o)] != n) { bad(1, "old_of_new / new_of_old are not inverse permutations at new id " + S(n)); break; }
    }
    if (P.n_tiles != int32_t((nel + kTile - 1) / kTile) || P.n_slices != P.n_tiles * (kTile / kSlice)) bad(2, "tile / slice counts do not cover the level");
    if (int64_t(P.slice_row0.size()) != P.n_slices + 1 || int64_t(P.rows_int.size()) != P.n_slices || int64_t(P.rows_bnd.size()) != P.n_slices) { bad(2, "slice tables have the wrong length"); return rep; }
    for (int32_t sl = 0; sl < P.n_slices; sl++)
        if (P.slice_row0[size_t(sl) + 1] - P.slice_row0[size_t(sl)] != P.rows_int[size_t(sl)] + P.rows_bnd[size_t(sl)] || P.rows_int[size_t(sl)] < 0 || P.rows_bnd[size_t(sl)] < 0)
            bad(3, "slice " + S(sl) + ": rows_int + rows_bnd != its share of slice_row0");
    const int64_t rows = P.slice_row0.back();
    if (int64_t(P.nbr16.size()) < rows * kSlice || int64_t(P.w.size()) < rows * kSlice || int64_t(P.nbr.size()) < rows * kSlice) { bad(3, "row arrays shorter than rows * 64"); return rep; }
    if (int64_t(P.tile_halo_ptr.size()) != P.n_tiles + 1 || int64_t(P.tile_ovf_ptr.size()) != P.n_tiles + 1) { bad(4, "tile halo tables have the wrong length"); return rep; }
    const bool tail_rows = P.has_tail;
    for (int32_t t = 0; t < P.n_tiles; t++) {
        const int64_t base = int64_t(t) * kTile;
        const int32_t n_here = int32_t(std::min<int64_t>(kTile, nel - base));
        const int32_t staged = P.tile_halo_ptr[size_t(t) + 1] - P.tile_halo_ptr[size_t(t)];
        const int32_t n_ovf = P.tile_ovf_ptr[size_t(t) + 1] - P.tile_ovf_ptr[size_t(t)];
        if (staged < 0 || staged > kHaloStride) bad(4, "tile " + S(t) + ": " + S(staged) + " staged halo nodes exceed the table's stride " + S(kHaloStride));
        if (staged > P.halo_max && n_ovf == 0) bad(4, "tile " + S(t) + ": halo_max understates its halo");
        for (int32_t k = P.tile_halo_ptr[size_t(t)]; k < P.tile_halo_ptr[size_t(t) + 1]; k++) {
            const int32_t id = P.tile_halo[size_t(k)];
            if (id < 0 || id >= nel) bad(5, "tile " + S(t) + ": halo id " + S(id) + " outside [0, nel)");
            else if (id >= base && id < base + kTile) bad(5, "tile " + S(t) + ": its own node " + S(id) + " listed as halo");
        }
        for (int32_t k = P.tile_ovf_ptr[size_t(t)]; k < P.tile_ovf_ptr[size_t(t) + 1]; k++)
            if (P.tile_ovf[size_t(k)] < 0 || P.tile_ovf[size_t(k)] >= nel) bad(5, "tile " + S(t) + ": overflow id outside [0, nel)");
        // what a 16-bit code may name, and that it names the node the 32-bit table names
        auto code_ok = [&](uint32_t c16, int32_t c32, const char *where) {
            const uint32_t slot = c16 & kT16SlotMask;
            if (slot == kT16Pad || slot == kT16Wall || slot == kT16Far) return;
            int64_t id = -1;
            if (slot < uint32_t(kTile)) { if (int32_t(slot) >= n_here) bad(6, std::string(where) + ": tile " + S(t) + " names own slot " + S(slot) + " of " + S(n_here)); id = base + slot; }
            else if (slot < uint32_t(kTile + staged)) id = P.tile_halo[size_t(P.tile_halo_ptr[size_t(t)]) + slot - kTile];
            else if (slot < uint32_t(kTileCap)) bad(6, std::string(where) + ": tile " + S(t) + " names LDS slot " + S(slot) + " beyond its " + S(staged) + " staged halo nodes");
            else if (int32_t(slot) - kTileCap >= n_ovf) bad(6, std::string(where) + ": tile " + S(t) + " names overflow entry " + S(int32_t(slot) - kTileCap) + " of " + S(n_ovf));
            else id = P.tile_ovf[size_t(P.tile_ovf_ptr[size_t(t)]) + slot - kTileCap];
            if (c32 >= 0 && id >= 0 && id != (c32 & kIdMask)) bad(7, std::string(where) + ": tile " + S(t) + ": the 16-bit code and the 32-bit table name different nodes");
        };
        const int32_t s0 = t * (kTile / kSlice);
        for (int32_t sl = s0; sl < s0 + kTile / kSlice; sl++) {
            for (int64_t row = P.slice_row0[size_t(sl)]; row < P.slice_row0[size_t(sl) + 1]; row++)
                for (int lane = 0; lane < kSlice; lane++) {
                    const size_t e = size_t(row) * kSlice + lane;
                    // (long rows: an entry moved to the workgroup's list is blanked in nbr16 and lives in tail_rec)
                    if (!(tail_rows && (P.nbr16[e] & kT16SlotMask) == kT16Pad)) code_ok(P.nbr16[e], P.nbr[e], "nbr16");
                }
        }
        if (P.has_tail) {
            const int32_t b = P.tail_tile_ptr[size_t(t)], e = P.tail_tile_ptr[size_t(t) + 1];
            if (b < 0 || e < b || e > P.tail_total || (b % 8) || (e % 8)) bad(8, "tile " + S(t) + ": list range [" + S(b) + ", " + S(e) + ") outside the list or not whole lines");
            for (int32_t k = b; k < e && k < P.tail_total; k++) {
                uint64_t word; std::memcpy(&word, &P.tail_rec[size_t(k) * 6 + 4], sizeof(word));
                const uint32_t own = uint32_t(word & 0xFFFFu), code = uint32_t(word >> 16) & 0xFFFFu;
                if (own == kT16Pad) continue;
                if (int32_t(own) >= n_here) bad(8, "tile " + S(t) + ": list entry owned by thread " + S(own) + " of " + S(n_here));
                code_ok(code, -1, "list entry");
            }
            for (int32_t k = 0; k < n_here; k++) {
                const int32_t nb = P.tail_begin[size_t(base + k)], nc = P.tail_count[size_t(base + k)];
                if (nc > 0 && (nb < b || nb + nc > e)) bad(8, "tile " + S(t) + ": a node's list entries leave its tile's range");
            }
        }
        if (P.free_rows) {
            const int32_t r0 = P.hr_row0[size_t(s0)], r1 = P.hr_row0[size_t(s0) + kTile / kSlice];
            if (P.half && r1 - r0 > kHalfTileRows) bad(9, "tile " + S(t) + ": " + S(r1 - r0) + " half rows exceed what the flux terms' LDS image holds");
            if (!P.free_wide && n_ovf != 0) bad(9, "tile " + S(t) + ": half rows on a tile with unstaged halo nodes and no table of their own");
            for (int32_t sl = s0; sl < s0 + kTile / kSlice; sl++) {
                const int32_t n_h = P.hr_row0[size_t(sl) + 1] - P.hr_row0[size_t(sl)];
                if (n_h < 0 || n_h > (P.half ? kHalfMaxRows : kFreeMaxRows)) bad(9, "slice " + S(sl) + ": " + S(n_h) + " half rows per lane (a lane keeps " + S(kHalfMaxRows) + ")");
                for (int64_t row = P.hr_row0[size_t(sl)]; row < P.hr_row0[size_t(sl) + 1]; row++)
                    for (int lane = 0; lane < kSlice; lane++) {
                        const uint32_t c = P.hr_code[size_t(row) * kSlice + lane];
                        if ((c & kT16SlotMask) == kT16Pad) continue;
                        if (P.free_wide) {
                            int32_t n_all = 0;
                            while (n_all < kFreeHaloStride && P.free_halo[size_t(t) * kFreeHaloStride + n_all] >= 0) n_all++;
                            const uint32_t sl_ = c & kT16SlotMask;
                            if (sl_ >= uint32_t(kTile + n_all) || (sl_ < uint32_t(kTile) && int32_t(sl_) >= n_here)) bad(10, "tile " + S(t) + ": a half row names slot " + S(sl_) + " beyond the " + S(n_all) + " halo nodes of its own table");
                        } else {
                            code_ok(c & 0xFFFFu, -1, "half row");
                            if ((c & kT16SlotMask) >= uint32_t(kTileCap)) bad(10, "half row names an overflow node");
                        }
                        const uint32_t own = (c >> 16) & 0xFFu, host = uint32_t((sl - s0) * kSlice + lane);
                        if (int32_t(own) >= n_here) bad(10, "tile " + S(t) + ": half row owned by thread " + S(own) + " of " + S(n_here));
                        if (((c & kHalfForeign) != 0) != (own != host)) bad(10, "tile " + S(t) + ": the foreign flag of a half row disagrees with its owner");
                        if ((c & kHalfMirror) && (c & kT16SlotMask) >= uint32_t(kTile)) bad(10, "tile " + S(t) + ": a half row hands -F to a node outside the tile");
                    }
                for (int64_t row = P.slice_row0[size_t(sl)]; row < P.slice_row0[size_t(sl)] + P.rows_int[size_t(sl)]; row++)
                    for (int lane = 0; lane < kSlice; lane++) {
                        const uint32_t g = P.hg16[size_t(row) * kSlice + lane] & kT16SlotMask;
                        if (g != kT16Pad && int32_t(g) >= (r1 - r0) * kSlice) bad(11, "tile " + S(t) + ": a gather position beyond the tile's flux terms");
                    }
            }
        }
        if (P.edge_once) {
            const int32_t n_te = P.te_count[size_t(t)];
            const int32_t chunks = P.te_chunk_ptr[size_t(t) + 1] - P.te_chunk_ptr[size_t(t)];
            if (n_te > chunks * kEdgeChunk || chunks > kMaxEdgeChunks) bad(12, "tile " + S(t) + ": " + S(n_te) + " listed edges in " + S(chunks) + " chunks");
            for (int32_t pp = 0; pp < n_te; pp++) {
                const size_t c = size_t(P.te_chunk_ptr[size_t(t)]) + size_t(pp / kEdgeChunk), ln = size_t(pp % kEdgeChunk);
                code_ok(P.te_slots[(c * 2 + 0) * kEdgeChunk + ln], -1, "edge list (a)");
                code_ok(P.te_slots[(c * 2 + 1) * kEdgeChunk + ln], -1, "edge list (b)");
            }
            for (int32_t sl = s0; sl < s0 + kTile / kSlice; sl++)
                for (int64_t row = P.slice_row0[size_t(sl)]; row < P.slice_row0[size_t(sl)] + P.rows_int[size_t(sl)]; row++)
                    for (int lane = 0; lane < kSlice; lane++) {
                        const uint32_t g = P.gat16[size_t(row) * kSlice + lane] & kT16SlotMask;
                        if (g != kT16Pad && int32_t(g) >= n_te) bad(12, "tile " + S(t) + ": a row refers to listed edge " + S(g) + " of " + S(n_te));
                    }
        }
    }
    // role 5 (k_flux_tile absorbing the first stage's time_step) reads old_variables / vin_flux / volumes at every staged node and
    // never the second halo slot or the overflow table: the solver launches it only under this condition (solver.cpp: vin_ok)
    const bool vin_ok = P.halo_overflow_refs == 0 && P.halo_max <= kTile;
    if (vin_ok)
        for (int32_t t = 0; t < P.n_tiles; t++)
            if (P.tile_halo_ptr[size_t(t) + 1] - P.tile_halo_ptr[size_t(t)] > kTile || P.tile_ovf_ptr[size_t(t) + 1] != P.tile_ovf_ptr[size_t(t)])
                bad(13, "tile " + S(t) + " violates what a role-5 launch assumes although the level claims it (vin_ok)");
    if (!P.child_ptr.empty()) {
        if (nel_coarse < 0 || int64_t(P.child_ptr.size()) != nel_coarse + 1) bad(14, "child_ptr is not nel_coarse + 1 long");
        else {
            for (int64_t c = 0; c < nel_coarse; c++) if (P.child_ptr[size_t(c) + 1] < P.child_ptr[size_t(c)]) bad(14, "child_ptr decreases at coarse node " + S(c));
            if (P.child_ptr.back() != int32_t(P.child.size())) bad(14, "child_ptr does not end at the number of children");
        }
        for (int32_t c : P.child) if (c < 0 || c >= nel) { bad(14, "a child id outside the fine level"); break; }
        for (int64_t n = 0; n < nel && n < int64_t(P.pro_parent.size()); n++) {
            const int32_t pr = P.pro_parent[size_t(n)], id = pr < 0 ? ~pr : pr;
            if (id < 0 || id >= nel_coarse) { bad(15, "node " + S(n) + ": parent outside the coarse level"); break; }
        }
        for (size_t e = 0; e < P.pro.size(); e++)
            if ((P.pro[e].w_own != 0.0 || P.pro[e].w_other != 0.0) && (P.pro[e].p_other < 0 || P.pro[e].p_other >= nel_coarse || P.pro[e].p_own < 0 || P.pro[e].p_own >= nel_coarse)) { bad(15, "a prolongation entry's parent outside the coarse level"); break; }
        if (P.pro_tiled)
            for (int32_t t = 0; t < P.n_tiles; t++) {
                const int32_t n_ids = P.pro_tile_n[size_t(t)];
                if (n_ids < 0 || n_ids > kProCap) bad(16, "tile " + S(t) + ": " + S(n_ids) + " staged coarse nodes");
                for (int32_t k = 0; k < n_ids; k++) if (P.pro_tile_ids[size_t(t) * kProCap + k] < 0 || P.pro_tile_ids[size_t(t) * kProCap + k] >= nel_coarse) bad(16, "tile " + S(t) + ": a staged coarse id outside the coarse level");
                const int64_t n0 = int64_t(t) * kTile, n1 = std::min<int64_t>(nel, n0 + kTile);
                for (int64_t n = n0; n < n1; n++) if (P.pro_own16[size_t(n)] >= n_ids) bad(16, "node " + S(n) + ": own parent's position beyond the tile's list");
                for (int32_t sl = t * (kTile / kSlice); sl < (t + 1) * (kTile / kSlice); sl++)
                    for (int64_t e = int64_t(P.slice_row0[size_t(sl)]) * kSlice; e < (int64_t(P.slice_row0[size_t(sl)]) + P.rows_int[size_t(sl)]) * kSlice; e++)
                        if ((P.pro[size_t(e)].w_own != 0.0 || P.pro[size_t(e)].w_other != 0.0) && P.pro_s16[size_t(e)] >= n_ids) bad(16, "tile " + S(t) + ": an entry's parent position beyond the tile's list");
            }
    }
    return rep;
}

} // namespace mgcfd
