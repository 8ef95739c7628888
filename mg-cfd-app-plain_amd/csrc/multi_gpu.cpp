// multi_gpu.cpp — euler3d_gpu_double --gpus N: the drop-in driver on N GPUs of one node, one process, written against
// the C ABI of include/mgcfd.h only (like main.cpp).
//
//   single-level input   the level is split over the N GPUs by recursive coordinate bisection (BASELINE configs[4]);
//                        every GPU owns its nodes and keeps read-only ghosts of the neighbours' nodes its edges touch;
//                        the sweep loop runs inside the library (mgcfd_group_sweeps: one all-reduce(MIN) of the time step
//                        per sweep, a halo message per neighbour after every Runge-Kutta stage, hidden under the
//                        interior tiles).  Owned nodes equal the one-GPU run bit for bit.
//   multigrid input      N <= levels: level l lives on GPU l % N (BASELINE configs[3]); the restricted variables go up and the
//                        coarse residuals come down as whole-array device-to-device copies (hipMemcpyPeerAsync over xGMI).
//                        The V-cycle is sequential in levels: this is placement, not concurrency.
//                        N > levels (or --gpus-partition): EVERY level is split over the N GPUs — level 0 by recursive
//                        coordinate bisection, a coarse node with its first child — and the whole V-cycle runs inside the
//                        library (mgcfd_group_cycles): the partitioned sweeps on every level, halo messages of the coarse
//                        variables after mg_restrict, of the coarse residuals before the prolongation and of the fine
//                        variables after it.  Every level equals the one-GPU run bit for bit on owned nodes.
//
// The reference has no multi-device path; its cycle loop (src/euler3d_cpu_double.cpp:371-694) fixes what has to move.
#include "multi_gpu.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <numeric>
#include <stdexcept>

#include <hip/hip_runtime_api.h>

namespace multi_gpu {

namespace {

struct Part {
    int64_t n_owned = 0;
    std::vector<int64_t> gids;                        // global id of every local node: owned (ascending), then ghosts (ascending)
    std::vector<double> volumes, coords;
    std::vector<mgcfd_edge> edges;                    // internal | boundary | wall, the whole mesh's relative order kept
    int64_t ni = 0, nb = 0, nw = 0;
    std::map<int, std::vector<int64_t>> send, recv;   // peer -> local ids (ascending global id on both sides)
    std::vector<int64_t> mg_map;                      // partitioned hierarchy: LOCAL coarse id of every local node's parent
};

// recursive coordinate bisection: split along the longest axis of the bounding box into halves whose sizes are
// proportional to the parts each will hold (equal counts to within a node, any n >= 1)
void rcb(const double *coords, std::vector<int64_t> &ids, int64_t b, int64_t e, int first, int count, std::vector<int> &part)
{
    if (count == 1 || e <= b) { for (int64_t k = b; k < e; k++) part[static_cast<size_t>(ids[static_cast<size_t>(k)])] = first; return; }
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int64_t k = b; k < e; k++)
        for (int d = 0; d < 3; d++) {
            const double c = coords[3 * ids[static_cast<size_t>(k)] + d];
            lo[d] = std::min(lo[d], c); hi[d] = std::max(hi[d], c);
        }
    int axis = 0;
    for (int d = 1; d < 3; d++) if (hi[d] - lo[d] > hi[axis] - lo[axis]) axis = d;
    std::stable_sort(ids.begin() + b, ids.begin() + e, [&](int64_t x, int64_t y) { return coords[3 * x + axis] < coords[3 * y + axis]; });
    const int left = count / 2;
    const int64_t cut = b + ((e - b) * left) / count;
    rcb(coords, ids, b, cut, first, left, part);
    rcb(coords, ids, cut, e, first + left, count - left, part);
}

std::vector<Part> partition_level(const mgcfd_level_desc &L, const std::vector<int> &part, int n_parts)
{
    std::vector<Part> parts(static_cast<size_t>(n_parts));
    const mgcfd_edge *E = L.edges;
    std::vector<int64_t> local(static_cast<size_t>(L.nel));
    for (int r = 0; r < n_parts; r++) {
        Part &P = parts[static_cast<size_t>(r)];
        std::vector<char> touched(static_cast<size_t>(L.nel), 0);
        for (int64_t i = 0; i < L.nel; i++) if (part[static_cast<size_t>(i)] == r) P.gids.push_back(i);
        P.n_owned = static_cast<int64_t>(P.gids.size());
        std::vector<int64_t> keep;
        for (int64_t e = L.internal_start; e < L.internal_start + L.n_internal; e++)
            if (part[static_cast<size_t>(E[e].a)] == r || part[static_cast<size_t>(E[e].b)] == r) {
                keep.push_back(e);
                touched[static_cast<size_t>(E[e].a)] = touched[static_cast<size_t>(E[e].b)] = 1;
            }
        P.ni = static_cast<int64_t>(keep.size());
        for (int64_t e = L.boundary_start; e < L.boundary_start + L.n_boundary; e++) if (part[static_cast<size_t>(E[e].b)] == r) { keep.push_back(e); P.nb++; }
        for (int64_t e = L.wall_start; e < L.wall_start + L.n_wall; e++) if (part[static_cast<size_t>(E[e].b)] == r) { keep.push_back(e); P.nw++; }
        for (int64_t i = 0; i < L.nel; i++) if (touched[static_cast<size_t>(i)] && part[static_cast<size_t>(i)] != r) P.gids.push_back(i);   // ghosts, ascending
        std::fill(local.begin(), local.end(), int64_t(-1));
        for (size_t k = 0; k < P.gids.size(); k++) local[static_cast<size_t>(P.gids[k])] = static_cast<int64_t>(k);
        for (int64_t g : P.gids) {
            P.volumes.push_back(L.volumes[g]);
            if (L.coords) for (int d = 0; d < 3; d++) P.coords.push_back(L.coords[3 * g + d]);
        }
        for (size_t k = 0; k < keep.size(); k++) {
            mgcfd_edge e = E[keep[k]];
            if (static_cast<int64_t>(k) < P.ni) e.a = local[static_cast<size_t>(e.a)];      // boundary / far-field faces keep their code in a
            e.b = local[static_cast<size_t>(e.b)];
            P.edges.push_back(e);
        }
        for (size_t k = static_cast<size_t>(P.n_owned); k < P.gids.size(); k++)
            P.recv[part[static_cast<size_t>(P.gids[k])]].push_back(static_cast<int64_t>(k));
    }
    // what a peer receives from me, in the same (ascending global id) order
    for (int r = 0; r < n_parts; r++)
        for (auto &kv : parts[static_cast<size_t>(r)].recv) {
            Part &Q = parts[static_cast<size_t>(kv.first)];
            std::vector<int64_t> &out = Q.send[r];
            for (int64_t lg : kv.second) {
                const int64_t g = parts[static_cast<size_t>(r)].gids[static_cast<size_t>(lg)];
                const auto it = std::lower_bound(Q.gids.begin(), Q.gids.begin() + Q.n_owned, g);
                out.push_back(static_cast<int64_t>(it - Q.gids.begin()));
            }
        }
    return parts;
}

// Every level of a hierarchy split over n_parts ranks (the C++ twin of mgcfd/partition.py: partition_hierarchy).  Level 0
// follows part0; a coarse node goes to the rank that owns its first child (a childless one to rank 0).  Besides the flux
// ghosts a rank holds, per level, what the transfers need: the children of its owned coarse nodes (mg_restrict computes a
// coarse node where it is owned, src/Kernels/mg_loops.cpp:30-202) and the parent of every local fine node (the prolongation
// reads the parents of a node's neighbours, mg_loops.cpp:678-864; the local map must be total).  Local edge lists keep the
// whole mesh's order, and a coarse node's children are summed by GLOBAL id (the gids go to the library as order keys).
// out[rank][level].
std::vector<std::vector<Part>> partition_hierarchy(const std::vector<mgcfd_level_desc> &L, const std::vector<int> &part0, int n_parts)
{
    const int n = static_cast<int>(L.size());
    // (what the library's create path checks as well: a map as long as its level, every entry a node of the coarser level —
    //  a shorter, longer or out-of-range map would index outside `owner` below; round 3's advisor finding)
    if (n == 0 || static_cast<int64_t>(part0.size()) != L[0].nel) throw std::invalid_argument("partition_hierarchy: one part per node of level 0 expected");
    for (int p : part0) if (p < 0 || p >= n_parts) throw std::invalid_argument("partition_hierarchy: a part index outside [0, n_parts)");
    for (int l = 0; l + 1 < n; l++) {
        const mgcfd_level_desc &D = L[static_cast<size_t>(l)];
        if (!D.mg_map || D.mgc != D.nel) throw std::invalid_argument("partition_hierarchy: the multigrid map of level " + std::to_string(l) + " must have one entry per node");
        const int64_t nc = L[static_cast<size_t>(l) + 1].nel;
        for (int64_t i = 0; i < D.mgc; i++)
            if (D.mg_map[i] < 0 || D.mg_map[i] >= nc) throw std::invalid_argument("partition_hierarchy: the multigrid map of level " + std::to_string(l) + " names a node outside the coarser level");
    }
    std::vector<std::vector<int>> owner(static_cast<size_t>(n));
    owner[0] = part0;
    for (int l = 0; l + 1 < n; l++) {
        const int64_t nc = L[static_cast<size_t>(l) + 1].nel;
        std::vector<int> o(static_cast<size_t>(nc), -1);
        for (int64_t i = 0; i < L[static_cast<size_t>(l)].mgc; i++) {
            const int64_t c = L[static_cast<size_t>(l)].mg_map[i];
            if (o[static_cast<size_t>(c)] < 0) o[static_cast<size_t>(c)] = owner[static_cast<size_t>(l)][static_cast<size_t>(i)];   // (ascending i: the first child)
        }
        for (int &v : o) if (v < 0) v = 0;
        owner[static_cast<size_t>(l) + 1] = std::move(o);
    }
    std::vector<std::vector<Part>> out(static_cast<size_t>(n_parts), std::vector<Part>(static_cast<size_t>(n)));
    for (int r = 0; r < n_parts; r++) {
        std::vector<std::vector<char>> local(static_cast<size_t>(n));
        for (int l = 0; l < n; l++) {
            const mgcfd_level_desc &D = L[static_cast<size_t>(l)];
            const std::vector<int> &own = owner[static_cast<size_t>(l)];
            std::vector<char> &in = local[static_cast<size_t>(l)];
            in.assign(static_cast<size_t>(D.nel), 0);
            for (int64_t i = 0; i < D.nel; i++) if (own[static_cast<size_t>(i)] == r) in[static_cast<size_t>(i)] = 1;
            for (int64_t e = D.internal_start; e < D.internal_start + D.n_internal; e++) {
                const int64_t a = D.edges[e].a, b = D.edges[e].b;
                if (own[static_cast<size_t>(a)] == r || own[static_cast<size_t>(b)] == r) in[static_cast<size_t>(a)] = in[static_cast<size_t>(b)] = 1;
            }
            if (l + 1 < n)                             // the children of the coarse nodes this rank owns
                for (int64_t i = 0; i < D.mgc; i++) if (owner[static_cast<size_t>(l) + 1][static_cast<size_t>(D.mg_map[i])] == r) in[static_cast<size_t>(i)] = 1;
            if (l > 0) {                               // the parent of every local node of the finer level
                const mgcfd_level_desc &F = L[static_cast<size_t>(l) - 1];
                const std::vector<char> &fin = local[static_cast<size_t>(l) - 1];
                for (int64_t i = 0; i < F.mgc; i++) if (fin[static_cast<size_t>(i)]) in[static_cast<size_t>(F.mg_map[i])] = 1;
            }
        }
        std::vector<std::vector<int64_t>> to_local(static_cast<size_t>(n));
        for (int l = 0; l < n; l++) {
            const mgcfd_level_desc &D = L[static_cast<size_t>(l)];
            const std::vector<int> &own = owner[static_cast<size_t>(l)];
            Part &P = out[static_cast<size_t>(r)][static_cast<size_t>(l)];
            for (int64_t i = 0; i < D.nel; i++) if (own[static_cast<size_t>(i)] == r) P.gids.push_back(i);
            P.n_owned = static_cast<int64_t>(P.gids.size());
            for (int64_t i = 0; i < D.nel; i++) if (local[static_cast<size_t>(l)][static_cast<size_t>(i)] && own[static_cast<size_t>(i)] != r) P.gids.push_back(i);
            std::vector<int64_t> &loc = to_local[static_cast<size_t>(l)];
            loc.assign(static_cast<size_t>(D.nel), int64_t(-1));
            for (size_t k = 0; k < P.gids.size(); k++) loc[static_cast<size_t>(P.gids[k])] = static_cast<int64_t>(k);
            for (int64_t g : P.gids) {
                P.volumes.push_back(D.volumes[g]);
                if (D.coords) for (int d = 0; d < 3; d++) P.coords.push_back(D.coords[3 * g + d]);
            }
            for (int64_t e = D.internal_start; e < D.internal_start + D.n_internal; e++)
                if (own[static_cast<size_t>(D.edges[e].a)] == r || own[static_cast<size_t>(D.edges[e].b)] == r) {
                    mgcfd_edge x = D.edges[e];
                    x.a = loc[static_cast<size_t>(x.a)]; x.b = loc[static_cast<size_t>(x.b)];
                    P.edges.push_back(x); P.ni++;
                }
            for (int64_t e = D.boundary_start; e < D.boundary_start + D.n_boundary; e++)
                if (own[static_cast<size_t>(D.edges[e].b)] == r) { mgcfd_edge x = D.edges[e]; x.b = loc[static_cast<size_t>(x.b)]; P.edges.push_back(x); P.nb++; }
            for (int64_t e = D.wall_start; e < D.wall_start + D.n_wall; e++)
                if (own[static_cast<size_t>(D.edges[e].b)] == r) { mgcfd_edge x = D.edges[e]; x.b = loc[static_cast<size_t>(x.b)]; P.edges.push_back(x); P.nw++; }
            for (size_t k = static_cast<size_t>(P.n_owned); k < P.gids.size(); k++) P.recv[own[static_cast<size_t>(P.gids[k])]].push_back(static_cast<int64_t>(k));
        }
        for (int l = 0; l + 1 < n; l++) {
            Part &P = out[static_cast<size_t>(r)][static_cast<size_t>(l)];
            for (int64_t g : P.gids) {
                const int64_t c = to_local[static_cast<size_t>(l) + 1][static_cast<size_t>(L[static_cast<size_t>(l)].mg_map[g])];
                if (c < 0) throw std::logic_error("partition_hierarchy: a local node's parent is not local");
                P.mg_map.push_back(c);
            }
        }
    }
    for (int l = 0; l < n; l++)
        for (int r = 0; r < n_parts; r++)
            for (auto &kv : out[static_cast<size_t>(r)][static_cast<size_t>(l)].recv) {
                Part &Q = out[static_cast<size_t>(kv.first)][static_cast<size_t>(l)];
                std::vector<int64_t> &snd = Q.send[r];
                for (int64_t lg : kv.second) {
                    const int64_t g = out[static_cast<size_t>(r)][static_cast<size_t>(l)].gids[static_cast<size_t>(lg)];
                    const auto it = std::lower_bound(Q.gids.begin(), Q.gids.begin() + Q.n_owned, g);
                    if (it == Q.gids.begin() + Q.n_owned || *it != g) throw std::logic_error("partition_hierarchy: a ghost's owner does not own it");
                    snd.push_back(static_cast<int64_t>(it - Q.gids.begin()));
                }
            }
    return out;
}

void set_halo(mgcfd_solver *s, int level, Part &P)
{
    std::vector<int> peers;
    for (auto &kv : P.send) peers.push_back(kv.first);
    for (auto &kv : P.recv) if (!P.send.count(kv.first)) peers.push_back(kv.first);
    std::sort(peers.begin(), peers.end());
    std::vector<int64_t> sc, rc;
    std::vector<const int64_t *> sp, rp;
    static const int64_t none = 0;
    for (int q : peers) {
        const auto &sv = P.send[q], &rv = P.recv[q];
        sc.push_back(static_cast<int64_t>(sv.size())); rc.push_back(static_cast<int64_t>(rv.size()));
        sp.push_back(sv.empty() ? &none : sv.data()); rp.push_back(rv.empty() ? &none : rv.data());
    }
    if (mgcfd_rank_set_halo(s, level, static_cast<int>(peers.size()), peers.data(), sc.data(), sp.data(), rc.data(), rp.data()) != MGCFD_OK)
        throw std::runtime_error(std::string("setting a rank's halo lists: ") + mgcfd_last_error());
}

void check(int rc, const char *what)
{
    if (rc != MGCFD_OK) throw std::runtime_error(std::string(what) + ": " + mgcfd_last_error());
}

int device_of_rank(const Options &o, int r) { return o.share_device ? o.first_device : o.first_device + r; }

} // namespace

struct Run::Impl {
    Options opt;
    int levels = 0, mesh_variant = 0;
    bool partitioned = false;
    bool partitioned_mg = false;                      // every level of a multigrid input split over the ranks (mgcfd_group_cycles)
    std::vector<std::vector<Part>> hparts;            // ... its parts, [rank][level] (parts = the level-0 parts then)
    std::vector<mgcfd_solver *> solvers;
    mgcfd_group *group = nullptr;
    std::vector<Part> parts;
    int64_t nel0 = 0;
    std::vector<int64_t> n_internal, nel;             // whole-mesh sizes per level (the reference's loop counts)
    std::vector<hipStream_t> streams;
    // level-per-GPU: staging arrays for the restricted variables
    std::vector<void *> stage;
};

Run::Run(const mgcfd_mesh *mesh, const Options &opt) : p(new Impl)
{
    p->opt = opt;
    p->levels = mgcfd_mesh_num_levels(mesh);
    p->mesh_variant = mgcfd_mesh_variant(mesh);
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess) have = 0;
    if (!opt.share_device && opt.first_device + opt.gpus > have)
        throw std::runtime_error("--gpus " + std::to_string(opt.gpus) + " but only " + std::to_string(have) + " GPU(s) are visible");
    for (int l = 0; l < p->levels; l++) {
        mgcfd_level_desc d;
        check(mgcfd_mesh_level(mesh, l, &d), "reading a level");
        p->nel.push_back(d.nel);
        p->n_internal.push_back(d.n_internal);
    }
    p->nel0 = p->nel[0];
    p->partitioned = p->levels == 1;
    p->partitioned_mg = p->levels > 1 && opt.gpus > 1 && (opt.gpus > p->levels || opt.partition_levels);
    if (p->partitioned_mg) {
        std::vector<mgcfd_level_desc> L(static_cast<size_t>(p->levels));
        for (int l = 0; l < p->levels; l++) check(mgcfd_mesh_level(mesh, l, &L[static_cast<size_t>(l)]), "reading a level");
        std::vector<int> part(static_cast<size_t>(L[0].nel), 0);
        if (L[0].coords) {
            std::vector<int64_t> ids(static_cast<size_t>(L[0].nel));
            std::iota(ids.begin(), ids.end(), int64_t(0));
            rcb(L[0].coords, ids, 0, L[0].nel, 0, opt.gpus, part);
        } else {
            for (int64_t i = 0; i < L[0].nel; i++) part[static_cast<size_t>(i)] = static_cast<int>((i * opt.gpus) / L[0].nel);
        }
        p->hparts = partition_hierarchy(L, part, opt.gpus);
        for (int r = 0; r < opt.gpus; r++) {
            std::vector<Part> &H = p->hparts[static_cast<size_t>(r)];
            std::vector<mgcfd_level_desc> d(static_cast<size_t>(p->levels));
            std::vector<int64_t> owned;
            std::vector<const int64_t *> keys;
            for (int l = 0; l < p->levels; l++) {
                Part &P = H[static_cast<size_t>(l)];
                mgcfd_level_desc &x = d[static_cast<size_t>(l)];
                x = mgcfd_level_desc{};
                x.nel = static_cast<int64_t>(P.gids.size());
                x.n_edges = static_cast<int64_t>(P.edges.size());
                x.n_internal = P.ni; x.n_boundary = P.nb; x.n_wall = P.nw;
                x.internal_start = 0; x.boundary_start = P.ni; x.wall_start = P.ni + P.nb;
                x.volumes = P.volumes.data();
                x.coords = P.coords.empty() ? nullptr : P.coords.data();
                x.edges = P.edges.data();
                x.mg_map = l + 1 < p->levels ? P.mg_map.data() : nullptr;
                x.mgc = l + 1 < p->levels ? static_cast<int64_t>(P.mg_map.size()) : 0;
                owned.push_back(P.n_owned);
                keys.push_back(P.gids.data());
            }
            mgcfd_solver *s = nullptr;
            check(mgcfd_create_partitioned_mg(d.data(), p->levels, p->mesh_variant, device_of_rank(opt, r), owned.data(), keys.data(), &s), "creating a rank's solver");
            p->solvers.push_back(s);
            p->parts.push_back(H[0]);                            // (get_level0 / check_invalid gather level 0 through these)
        }
        check(mgcfd_group_create(opt.gpus, p->solvers.data(), &p->group), "forming the group");
        for (int r = 0; r < opt.gpus; r++)
            for (int l = 0; l < p->levels; l++) set_halo(p->solvers[static_cast<size_t>(r)], l, p->hparts[static_cast<size_t>(r)][static_cast<size_t>(l)]);
        for (int l = 0; l < p->levels; l++) check(mgcfd_group_exchange(p->group, l), "the first halo exchange");
    } else if (p->partitioned) {
        mgcfd_level_desc L;
        check(mgcfd_mesh_level(mesh, 0, &L), "reading level 0");
        std::vector<int> part(static_cast<size_t>(L.nel), 0);
        if (L.coords) {
            std::vector<int64_t> ids(static_cast<size_t>(L.nel));
            std::iota(ids.begin(), ids.end(), int64_t(0));
            rcb(L.coords, ids, 0, L.nel, 0, opt.gpus, part);
        } else {
            // no coordinates (a single-level fvcorr input has no .coords file): equal ranges of node ids — any
            // partition gives the same results, a good one fewer ghosts
            for (int64_t i = 0; i < L.nel; i++) part[static_cast<size_t>(i)] = static_cast<int>((i * opt.gpus) / L.nel);
        }
        p->parts = partition_level(L, part, opt.gpus);
        for (int r = 0; r < opt.gpus; r++) {
            Part &P = p->parts[static_cast<size_t>(r)];
            mgcfd_level_desc d{};
            d.nel = static_cast<int64_t>(P.gids.size());
            d.n_edges = static_cast<int64_t>(P.edges.size());
            d.n_internal = P.ni; d.n_boundary = P.nb; d.n_wall = P.nw;
            d.internal_start = 0; d.boundary_start = P.ni; d.wall_start = P.ni + P.nb;
            d.volumes = P.volumes.data();
            d.coords = P.coords.empty() ? nullptr : P.coords.data();
            d.edges = P.edges.data();
            mgcfd_solver *s = nullptr;
            check(mgcfd_create_partitioned(&d, 1, p->mesh_variant, device_of_rank(opt, r), &P.n_owned, &s), "creating a rank's solver");
            p->solvers.push_back(s);
        }
        check(mgcfd_group_create(opt.gpus, p->solvers.data(), &p->group), "forming the group");
        for (int r = 0; r < opt.gpus; r++) set_halo(p->solvers[static_cast<size_t>(r)], 0, p->parts[static_cast<size_t>(r)]);
        check(mgcfd_group_exchange(p->group, 0), "the first halo exchange");
    } else {
        // one multigrid level per GPU: every rank holds the hierarchy (plans and static data), sweeps only its levels
        for (int r = 0; r < std::min(opt.gpus, p->levels); r++) {
            mgcfd_solver *s = nullptr;
            check(mgcfd_create_from_mesh(mesh, device_of_rank(opt, r), &s), "creating a rank's solver");
            p->solvers.push_back(s);
        }
        p->stage.assign(static_cast<size_t>(p->levels), nullptr);
    }
    for (mgcfd_solver *s : p->solvers) {
        mgcfd_set_option(s, MGCFD_OPT_EXACT, opt.fast_math ? 0 : 1);
        mgcfd_set_option(s, MGCFD_OPT_TIMING, 0);                 // the fused path: per-loop times are not collected across devices
    }
}

Run::~Run()
{
    if (p->group) mgcfd_group_destroy(p->group);
    for (size_t l = 0; l < p->stage.size(); l++) if (p->stage[l]) (void)hipFree(p->stage[l]);
    for (mgcfd_solver *s : p->solvers) mgcfd_destroy(s);
    delete p;
}

int Run::ranks() const { return static_cast<int>(p->solvers.size()); }
bool Run::partitioned() const { return p->partitioned || p->partitioned_mg; }
bool Run::partitioned_hierarchy() const { return p->partitioned_mg; }

// level-per-GPU: move a whole node array of `level` from the solver of rank `src` to the solver of rank `dst`
static void hand_over(Run::Impl *p, int level, int which, int src, int dst, bool restricted)
{
    if (src == dst) return;
    mgcfd_solver *a = p->solvers[static_cast<size_t>(src)], *b = p->solvers[static_cast<size_t>(dst)];
    void *from = nullptr, *to = nullptr;
    int64_t count = 0, count_b = 0;
    check(mgcfd_array_devptr(a, level, which, &from, &count), "array address");
    check(mgcfd_array_devptr(b, level, which, &to, &count_b), "array address");
    check(mgcfd_synchronize(a), "synchronising the sender");     // (placement, not concurrency: the levels run one after another anyway)
    const int da = device_of_rank(p->opt, src), db = device_of_rank(p->opt, dst);
    if (restricted) {
        // mg_restrict leaves a coarse node without children at ITS old value, which only the receiving rank has
        if (!p->stage[static_cast<size_t>(level)]) {
            if (hipSetDevice(db) != hipSuccess || hipMalloc(&p->stage[static_cast<size_t>(level)], sizeof(double) * static_cast<size_t>(count)) != hipSuccess)
                throw std::runtime_error("allocating a staging array");
        }
        if (hipMemcpyPeer(p->stage[static_cast<size_t>(level)], db, from, da, sizeof(double) * static_cast<size_t>(count)) != hipSuccess)
            throw std::runtime_error("device-to-device copy failed");
        check(mgcfd_accept_restricted(b, level - 1, p->stage[static_cast<size_t>(level)]), "taking the restricted variables");
    } else {
        check(mgcfd_synchronize(b), "synchronising the receiver");
        if (hipMemcpyPeer(to, db, from, da, sizeof(double) * static_cast<size_t>(count)) != hipSuccess)
            throw std::runtime_error("device-to-device copy failed");
        check(mgcfd_array_written(b, level, which), "marking the array written");
    }
}

int Run::run_cycles(int cycles, double *rms_out)
{
    const int n = p->levels, w = ranks();
    if (p->partitioned_mg) {
        // every level partitioned: the whole batch of V-cycles inside the library, the RMS of every cycle read back once
        std::vector<double> rms(static_cast<size_t>(std::max(cycles, 1)));
        for (int c = 0; c < cycles; c += 4096) {
            const int rc = mgcfd_group_cycles(p->group, std::min(4096, cycles - c), rms.data() + c);
            if (rc == MGCFD_ERR_NAN || rc == MGCFD_ERR_NEG_DENSITY || rc == MGCFD_ERR_NEG_ENERGY) return rc;
            check(rc, "the partitioned V-cycles");
        }
        if (rms_out) std::copy(rms.begin(), rms.begin() + cycles, rms_out);
        return MGCFD_OK;
    }
    if (p->partitioned) {
        // the whole batch inside the library: a host thread per rank, the RMS of every cycle read back once
        std::vector<double> rms(static_cast<size_t>(std::max(cycles, 1)));
        for (int c = 0; c < cycles; c += 4096) {
            const int n_now = std::min(4096, cycles - c);
            check(mgcfd_group_sweeps_rms(p->group, 0, n_now, rms.data() + c), "the partitioned sweeps");
        }
        if (rms_out) std::copy(rms.begin(), rms.begin() + cycles, rms_out);
    }
    for (int c = 0; c < cycles && !p->partitioned; c++) {
        {
            auto owner = [&](int l) { return l % w; };
            for (int l = 0; l < n; l++) {
                mgcfd_solver *s = p->solvers[static_cast<size_t>(owner(l))];
                check(mgcfd_smooth(s, l, 1), "a sweep");
                if (l == 0) { double rms = 0.0; check(mgcfd_calc_rms(s, 0, &rms), "the RMS"); if (rms_out) rms_out[c] = rms; }
                if (l + 1 < n) {
                    check(mgcfd_restrict(s, l), "restrict");                 // fills level l+1's variables on the FINE level's rank
                    hand_over(p, l + 1, MGCFD_ARR_VARIABLES, owner(l), owner(l + 1), true);
                }
            }
            for (int l = n - 2; l >= 0; l--) {
                hand_over(p, l + 1, MGCFD_ARR_RESIDUALS, owner(l + 1), owner(l), false);
                mgcfd_solver *s = p->solvers[static_cast<size_t>(owner(l))];
                check(mgcfd_prolong(s, l), "prolong");
                if (l > 0) check(mgcfd_smooth(s, l, 1), "a sweep");
            }
        }
    }
    // check_for_invalid_variables (src/Kernels/validation.cpp:107-138): every fused stage carried the check
    for (int r = 0; r < w; r++)
        for (int l = 0; l < (p->partitioned ? 1 : n); l++) {
            if (!p->partitioned && l % w != r) continue;
            int64_t bad = -1;
            // (what the stages' own checks found: the reference checks after a time_step only — a value spoilt by the last
            //  prolongation is not an error there, src/euler3d_cpu_double.cpp:383-508)
            (void)l;
            const int rc = mgcfd_pending_invalid_state(p->solvers[static_cast<size_t>(r)], &bad);
            if (rc != MGCFD_OK) return rc;
        }
    return MGCFD_OK;
}

void Run::get_level0(int which, int ncols, double *out) const
{
    if (!p->partitioned && !p->partitioned_mg) { check(mgcfd_get_array(p->solvers[0], 0, which, out), "reading back an array"); return; }
    for (size_t r = 0; r < p->solvers.size(); r++) {
        const Part &P = p->parts[r];
        std::vector<double> a(P.gids.size() * static_cast<size_t>(ncols));
        check(mgcfd_get_array(p->solvers[r], 0, which, a.data()), "reading back an array");
        for (int64_t k = 0; k < P.n_owned; k++)
            std::memcpy(out + P.gids[static_cast<size_t>(k)] * ncols, a.data() + static_cast<size_t>(k) * ncols, sizeof(double) * static_cast<size_t>(ncols));
    }
}

int Run::check_invalid(int level, int64_t *bad_cell) const
{
    // (validation.cpp:107-138 stops at the first bad cell in original order: the smallest global id over the ranks)
    if (!p->partitioned && !p->partitioned_mg) return mgcfd_check_for_invalid_variables(p->solvers[static_cast<size_t>(level % ranks())], level, bad_cell);
    int rc_all = MGCFD_OK;
    int64_t first = -1;
    for (size_t r = 0; r < p->solvers.size(); r++) {
        int64_t bad = -1;
        const int lvl = p->partitioned_mg ? level : 0;
        const std::vector<int64_t> &gids = p->partitioned_mg ? p->hparts[r][static_cast<size_t>(level)].gids : p->parts[r].gids;
        const int rc = mgcfd_check_for_invalid_variables(p->solvers[r], lvl, &bad);
        if (rc == MGCFD_OK) continue;
        const int64_t g = (bad >= 0 && bad < int64_t(gids.size())) ? gids[static_cast<size_t>(bad)] : bad;
        if (first < 0 || g < first) { first = g; rc_all = rc; }
    }
    if (bad_cell) *bad_cell = first;
    return rc_all;
}

void Run::loop_iters(int level, int cycles, int64_t out[MGCFD_NUM_LOOPS]) const
{
    // what the reference's counters would hold for the whole mesh (src/Monitoring/loop_stats.cpp:48-81): a rank's own
    // counters include the edges cut by the partition once per side
    std::memset(out, 0, sizeof(int64_t) * MGCFD_NUM_LOOPS);
    if (p->partitioned_mg) {
        // (a V-cycle sweeps levels 0 .. n-1, n-2 .. 1: every level but the first and the last twice; restrict is booked on the
        //  coarse level, op_restrict / op_prolong of solver.cpp give the counts: mg_loops.cpp:61,117,172 and :728,842)
        const int n = p->levels;
        const int64_t sw = (level == 0 || level == n - 1) ? 1 : 2;
        out[MGCFD_LOOP_FLUX] = int64_t(MGCFD_RK) * sw * cycles * p->n_internal[static_cast<size_t>(level)];
        out[MGCFD_LOOP_COMPUTE_STEP] = sw * cycles * p->nel[static_cast<size_t>(level)];
        out[MGCFD_LOOP_TIME_STEP] = int64_t(MGCFD_RK) * sw * cycles * p->nel[static_cast<size_t>(level)];
        if (level > 0) out[MGCFD_LOOP_RESTRICT] = int64_t(cycles) * (2 * p->nel[static_cast<size_t>(level) - 1] + p->nel[static_cast<size_t>(level)]);
        if (level + 1 < n) out[MGCFD_LOOP_PROLONG] = int64_t(cycles) * (p->n_internal[static_cast<size_t>(level)] + p->nel[static_cast<size_t>(level)]);
        return;
    }
    if (p->partitioned) {
        out[MGCFD_LOOP_FLUX] = int64_t(MGCFD_RK) * cycles * p->n_internal[0];
        out[MGCFD_LOOP_COMPUTE_STEP] = int64_t(cycles) * p->nel[0];
        out[MGCFD_LOOP_TIME_STEP] = int64_t(MGCFD_RK) * cycles * p->nel[0];
        return;
    }
    check(mgcfd_get_loop_iters(p->solvers[static_cast<size_t>(level % ranks())], level, out), "loop counters");
    if (level > 0) {                                             // restrict is counted on the coarse level, where it ran on the fine level's rank
        int64_t fine[MGCFD_NUM_LOOPS];
        check(mgcfd_get_loop_iters(p->solvers[static_cast<size_t>((level - 1) % ranks())], level, fine), "loop counters");
        out[MGCFD_LOOP_RESTRICT] = fine[MGCFD_LOOP_RESTRICT];
    }
}

} // namespace multi_gpu
