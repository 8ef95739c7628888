// mesh.cpp — readers/writers for the reference's input and output formats.
// Behavioural contract (what must match the reference):
//   * input.dat grammar and error conditions      src/Base/io_enhanced.cpp:407-579
//   * mesh text file -> classified edge list      src/Base/io.cpp:56-177
//   * .coords only when levels > 1                src/Base/io.cpp:49-54,77-81
//   * MG map file                                 src/Base/io_enhanced.cpp:629-650
//   * -m duplication                              src/Base/io_enhanced.cpp:89-201
//   * "%.17e" dumps                               src/Base/io.cpp:201-233
#include "mesh.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <charconv>
#include <future>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>

namespace mgcfd {

namespace {

[[noreturn]] void fail(const std::string &msg) { throw std::runtime_error(msg); }

std::string trimmed(const std::string &s)
{
    size_t b = s.find_first_not_of(" \t\r\n");
    if (b == std::string::npos) return "";
    size_t e = s.find_last_not_of(" \t\r\n");
    return s.substr(b, e - b + 1);
}

// Whole-file whitespace tokenizer (the reference parses with operator>>, so any whitespace layout is legal).  The file is
// mapped, not copied; numbers are converted by std::from_chars — correctly rounded, like strtod, and several times faster
// (a 75 MB level file is most of the drop-in's start-up) — and anything from_chars does not take as strtod / strtol would
// (a leading '+', hexadecimal floats, a number that runs into the end of the mapping) goes through strtod / strtol on a copy
// of the token, so every file the old reader accepted parses to the same bits (tools/fuzz_reader.py).
class TokenFile {
public:
    explicit TokenFile(const std::string &path) : path_(path)
    {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) fail("could not open data file: '" + path + "'");
        struct stat st;
        if (::fstat(fd, &st) != 0) { ::close(fd); fail("could not open data file: '" + path + "'"); }
        size_ = static_cast<size_t>(st.st_size);
        if (size_ > 0) {
            void *m = ::mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) {
                // (not every file system maps: read it instead)
                own_.resize(size_);
                size_t got = 0;
                while (got < size_) {
                    const ssize_t r = ::read(fd, own_.data() + got, size_ - got);
                    if (r <= 0) break;
                    got += static_cast<size_t>(r);
                }
                size_ = got;
                base_ = own_.data();
            } else {
                map_ = m;
                base_ = static_cast<const char *>(m);
                ::madvise(m, size_, MADV_SEQUENTIAL);
            }
        }
        ::close(fd);
        cur_ = base_;
        end_ = base_ + size_;
    }
    ~TokenFile() { if (map_) ::munmap(map_, size_); }
    TokenFile(const TokenFile &) = delete;
    TokenFile &operator=(const TokenFile &) = delete;

    long next_long()
    {
        skip_space();
        long v = 0;
        if (cur_ < end_ && *cur_ != '+') {
            const auto r = std::from_chars(cur_, end_, v, 10);
            if (r.ec == std::errc() && r.ptr < end_) { cur_ = r.ptr; return v; }
        }
        return slow_long();
    }
    double next_double()
    {
        skip_space();
        double v = 0.0;
        if (cur_ < end_ && *cur_ != '+') {
            const auto r = std::from_chars(cur_, end_, v, std::chars_format::general);
            // (a token that strtod would read further — "0x..." stops from_chars behind the 0 — takes the slow path)
            if (r.ec == std::errc() && r.ptr < end_ && *r.ptr != 'x' && *r.ptr != 'X') { cur_ = r.ptr; return v; }
        }
        return slow_double();
    }
    // every number up to the first thing that is not one (or the end of the file)
    std::vector<double> all_doubles()
    {
        std::vector<double> out;
        out.reserve(size_ / 20);
        for (;;) {
            skip_space();
            if (cur_ >= end_) break;
            const char *before = cur_;
            double v;
            try { v = next_double(); } catch (const std::runtime_error &) { cur_ = before; break; }
            out.push_back(v);
        }
        return out;
    }
private:
    void skip_space()
    {
        // (the C locale's isspace set, what strtol / strtod skip)
        while (cur_ < end_ && (*cur_ == ' ' || (*cur_ >= '\t' && *cur_ <= '\r'))) cur_++;
    }
    // the token (up to 4 KB of it) as a C string: what strtol / strtod would have seen in the old reader's buffer
    std::string token_copy() const
    {
        const char *e = cur_;
        while (e < end_ && e - cur_ < 4096 && !(*e == ' ' || (*e >= '\t' && *e <= '\r'))) e++;
        return std::string(cur_, e);
    }
    long slow_long()
    {
        const std::string tok = token_copy();
        char *end = nullptr;
        errno = 0;
        const long v = std::strtol(tok.c_str(), &end, 10);
        if (end == tok.c_str()) fail("unexpected end of data / malformed integer in '" + path_ + "'");
        cur_ += end - tok.c_str();
        return v;
    }
    double slow_double()
    {
        const std::string tok = token_copy();
        char *end = nullptr;
        const double v = std::strtod(tok.c_str(), &end);
        if (end == tok.c_str()) fail("unexpected end of data / malformed number in '" + path_ + "'");
        cur_ += end - tok.c_str();
        return v;
    }
    std::string path_;
    std::vector<char> own_;
    void *map_ = nullptr;
    const char *base_ = nullptr, *cur_ = nullptr, *end_ = nullptr;
    size_t size_ = 0;
};

bool key_value(const std::string &line, std::string &key, std::string &value)
{
    size_t eq = line.find('=');
    if (eq == std::string::npos || eq + 1 > line.size()) return false;
    key = trimmed(line.substr(0, eq));
    value = trimmed(line.substr(eq + 1));
    // the reference ignores "key=" with nothing after it (second getline fails)
    return eq + 1 < line.size();
}

} // namespace

const char *mesh_variant_name(int v)
{
    switch (v) {
        case MGCFD_MESH_FVCORR: return "fvcorr";
        case MGCFD_MESH_M6_WING: return "m6wing";
        case MGCFD_MESH_LA_CASCADE: return "la_cascade";
        case MGCFD_MESH_ROTOR_37: return "rotor37";
        default: return "unknown";
    }
}

mgcfd_level_desc HostLevel::desc() const
{
    mgcfd_level_desc d;
    d.nel = nel;
    d.n_edges = static_cast<int64_t>(edges.size());
    d.n_internal = n_internal; d.n_boundary = n_boundary; d.n_wall = n_wall;
    d.internal_start = internal_start; d.boundary_start = boundary_start; d.wall_start = wall_start;
    d.volumes = volumes.data();
    d.coords = have_coords ? coords.data() : nullptr;
    d.edges = edges.data();
    d.mg_map = mg_map.empty() ? nullptr : mg_map.data();
    d.mgc = static_cast<int64_t>(mg_map.size());
    return d;
}

InputDat parse_input_dat(const std::string &path)
{
    std::ifstream file(path);
    if (!file.is_open()) fail("Error: Could not open input file '" + path + "'");
    InputDat in;
    bool have_size = false, have_levels = false, have_name = false, have_files = false;
    std::string line;
    auto read_section = [&](std::vector<std::string> &dst, int count, const char *what) {
        dst.assign(static_cast<size_t>(count > 0 ? count : 0), "");
        for (int i = 0; i < count; i++) {
            if (!std::getline(file, line))
                fail("Error parsing " + path + ": Have reached EOF before reading all " + what + " filenames");
            std::string k, v;
            if (line.empty())
                fail("Error parsing '" + path + "': Was expecting a key-value pair following [" +
                     std::string(what) + "]");
            if (key_value(line, k, v)) {
                int idx = std::atoi(k.c_str());
                if (idx < 0 || idx >= count) fail("Error parsing '" + path + "': level index out of range");
                dst[static_cast<size_t>(idx)] = v;
            }
        }
    };
    while (std::getline(file, line)) {
        if (!line.empty() && line[0] == '#') continue;
        if (!line.empty() && line[0] == '[') {
            // section headers must match exactly (strcmp in the reference)
            if (line == "[levels]" || line == "[mg_mapping]") {
                if (!have_levels)
                    fail("Error parsing " + path + ": Need to know number of levels before parsing level filenames");
                if (line == "[levels]") { read_section(in.level_files, in.num_levels, "levels"); have_files = true; }
                else read_section(in.map_files, in.num_levels - 1, "mg_mapping");
            }
            continue;
        }
        std::string k, v;
        if (!key_value(line, k, v)) continue;
        if (k == "size") { in.size = std::atoi(v.c_str()); have_size = true; }
        else if (k == "num_levels") { in.num_levels = std::atoi(v.c_str()); have_levels = true; }
        else if (k == "mesh_name") {
            if (v == "la_cascade") in.mesh_variant = MGCFD_MESH_LA_CASCADE;
            else if (v == "rotor37") in.mesh_variant = MGCFD_MESH_ROTOR_37;
            else if (v == "fvcorr") in.mesh_variant = MGCFD_MESH_FVCORR;
            else if (v == "m6wing") in.mesh_variant = MGCFD_MESH_M6_WING;
            else fail("Error parsing " + path + ": Unknown mesh_name '" + v + "'");
            in.mesh_name = v;
            have_name = true;
        }
    }
    if (!have_size) fail("Error parsing '" + path + "': size not present");
    if (!have_levels) fail("Error parsing '" + path + "': number of levels not present");
    if (!have_name) fail("Error parsing '" + path + "': mesh name not present");
    if (!have_files) fail("Error parsing '" + path + "': mesh filenames not present");
    if (in.map_files.empty() && in.num_levels > 1) in.map_files.assign(static_cast<size_t>(in.num_levels - 1), "");
    return in;
}

namespace {

// A level's three files are parsed by a task each (load_mesh); what a task found wrong is kept and raised in the order the
// serial reader would have met it: the level file's header, the coordinates, the level file's body, the multigrid map.
struct LevelParse {
    HostLevel L;
    std::exception_ptr header_error, body_error;
    std::string warning;
};

LevelParse parse_level_file(const std::string &path, int mesh_variant)
{
    LevelParse out;
    HostLevel &L = out.L;
    long declared_edges = 0;
    std::unique_ptr<TokenFile> fp;
    try {
        fp.reset(new TokenFile(path));
        L.nel = fp->next_long();
        declared_edges = fp->next_long();
        if (L.nel < 0 || declared_edges < 0) fail("negative size in '" + path + "'");
    } catch (...) { out.header_error = std::current_exception(); return out; }
    try {
        TokenFile &f = *fp;
        L.volumes.resize(static_cast<size_t>(L.nel));
        // Classify while reading; each class keeps file order.
        std::vector<mgcfd_edge> cls[3];
        cls[0].reserve(static_cast<size_t>(declared_edges));
        for (int64_t i = 0; i < L.nel; i++) {
            L.volumes[static_cast<size_t>(i)] = f.next_double();
            const long degree = f.next_long();
            for (long j = 0; j < degree; j++) {
                const long nb = f.next_long();
                double wx = f.next_double(), wy = f.next_double(), wz = f.next_double();
                if (nb >= i) continue;              // recorded once, from the higher-numbered end
                const int k = nb >= 0 ? 0 : (nb == -1 ? 1 : (nb == -2 ? 2 : 0));
                // Rodinia's fvcorr flips every normal; other meshes flip only the internal
                // edges, which are being recorded backwards (b -> a).
                if (mesh_variant == MGCFD_MESH_FVCORR || nb >= 0) { wx *= -1; wy *= -1; wz *= -1; }
                cls[k].push_back(mgcfd_edge{nb, i, wx, wy, wz});
            }
        }
        L.n_internal = static_cast<int64_t>(cls[0].size());
        L.n_boundary = static_cast<int64_t>(cls[1].size());
        L.n_wall = static_cast<int64_t>(cls[2].size());
        const int64_t found = L.n_internal + L.n_boundary + L.n_wall;
        if (found != declared_edges) {
            char buf[160];
            std::snprintf(buf, sizeof(buf), "WARNING: Mesh claims to have %ld edges, actually has %ld\n", declared_edges, (long)found);
            out.warning = buf;
        }
        if (found > declared_edges)
            fail("mesh '" + path + "' lists more edges than its header declares (the reference would overrun its buffers)");
        L.internal_start = 0;
        L.boundary_start = L.n_internal;
        L.wall_start = L.n_internal + L.n_boundary;
        L.edges.reserve(static_cast<size_t>(declared_edges));
        for (auto &c : cls) L.edges.insert(L.edges.end(), c.begin(), c.end());
        L.edges.resize(static_cast<size_t>(declared_edges), mgcfd_edge{-5, -5, 0.0, 0.0, 0.0});
    } catch (...) { out.body_error = std::current_exception(); }
    return out;
}

// the first `count` numbers of a coordinates file (missing file is fatal when levels > 1); `count` < 0: as many as the file
// holds, in threes (the level's size is in another file, parsed meanwhile)
std::vector<double> parse_coords_file(const std::string &path)
{
    TokenFile c(path);
    std::vector<double> out;
    return c.all_doubles();
}

// raise what the serial reader would have raised first, print what it would have printed
HostLevel finish_level(LevelParse &&p, std::future<std::vector<double>> *coords, const std::string &path)
{
    if (p.header_error) { if (coords) coords->wait(); std::rethrow_exception(p.header_error); }
    HostLevel L = std::move(p.L);
    L.coords.assign(static_cast<size_t>(L.nel) * 3, 0.0);
    if (coords) {
        std::vector<double> c = coords->get();                 // (raises the coordinate file's own error)
        if (static_cast<int64_t>(c.size()) < L.nel * 3) fail("unexpected end of data / malformed number in '" + path + ".coords'");
        std::copy(c.begin(), c.begin() + L.nel * 3, L.coords.begin());
        L.have_coords = true;
    }
    if (!p.warning.empty()) std::fputs(p.warning.c_str(), stdout);
    if (p.body_error) std::rethrow_exception(p.body_error);
    return L;
}

} // namespace

HostLevel read_mesh_level(const std::string &path, int mesh_variant, bool read_coords)
{
    std::future<std::vector<double>> coords;
    if (read_coords) coords = std::async(std::launch::async, parse_coords_file, path + ".coords");
    LevelParse p = parse_level_file(path, mesh_variant);
    return finish_level(std::move(p), read_coords ? &coords : nullptr, path);
}

std::vector<int64_t> read_mg_map(const std::string &path)
{
    std::FILE *probe = std::fopen(path.c_str(), "rb");
    if (!probe) fail("could not open mg file: '" + path + "'");
    std::fclose(probe);
    TokenFile f(path);
    const long n = f.next_long();
    if (n < 0) fail("negative mgc in '" + path + "'");
    std::vector<int64_t> map(static_cast<size_t>(n));
    for (long i = 0; i < n; i++) map[static_cast<size_t>(i)] = f.next_long();
    return map;
}

void duplicate_level(HostLevel &L, int m, int64_t nel_above)
{
    if (m <= 1) return;
    const int64_t nel = L.nel;
    std::vector<double> vol(static_cast<size_t>(nel * m)), crd(static_cast<size_t>(nel * m * 3));
    for (int c = 0; c < m; c++) {
        std::copy(L.volumes.begin(), L.volumes.end(), vol.begin() + c * nel);
        std::copy(L.coords.begin(), L.coords.end(), crd.begin() + c * nel * 3);
    }
    const int64_t declared = static_cast<int64_t>(L.edges.size());
    const int64_t starts[3] = {L.internal_start, L.boundary_start, L.wall_start};
    const int64_t counts[3] = {L.n_internal, L.n_boundary, L.n_wall};
    std::vector<mgcfd_edge> ed;
    ed.reserve(static_cast<size_t>(declared * m));
    const mgcfd_edge pad{-5, -5, 0.0, 0.0, 0.0};
    for (int k = 0; k < 3; k++) {
        ed.resize(static_cast<size_t>(starts[k] * m), pad);       // class k starts at m x its old start
        for (int c = 0; c < m; c++)
            for (int64_t e = 0; e < counts[k]; e++) {
                mgcfd_edge r = L.edges[static_cast<size_t>(starts[k] + e)];
                if (r.a >= 0) r.a += nel * c;
                if (r.b >= 0) r.b += nel * c;
                ed.push_back(r);
            }
    }
    ed.resize(static_cast<size_t>(declared * m), pad);
    if (!L.mg_map.empty()) {
        const int64_t mgc = static_cast<int64_t>(L.mg_map.size());
        std::vector<int64_t> map(static_cast<size_t>(mgc * m));
        for (int c = 0; c < m; c++)
            for (int64_t n = 0; n < mgc; n++) map[static_cast<size_t>(c * mgc + n)] = L.mg_map[static_cast<size_t>(n)] + nel_above * c;
        L.mg_map.swap(map);
    }
    L.volumes.swap(vol);
    L.coords.swap(crd);
    L.edges.swap(ed);
    L.nel *= m;
    L.n_internal *= m; L.n_boundary *= m; L.n_wall *= m;
    L.boundary_start *= m; L.wall_start *= m;
}

void sort_edges_legacy(HostLevel &L)
{
    auto before = [](const mgcfd_edge &p, const mgcfd_edge &q) {
        if (p.a != q.a) return p.a < q.a;
        if (p.b != q.b) return p.b < q.b;
        if (p.x != q.x) return p.x < q.x;
        if (p.y != q.y) return p.y < q.y;
        return p.z < q.z;
    };
    std::sort(L.edges.begin() + L.internal_start, L.edges.begin() + L.internal_start + L.n_internal, before);
    std::sort(L.edges.begin() + L.boundary_start, L.edges.begin() + L.boundary_start + L.n_boundary, before);
    std::sort(L.edges.begin() + L.wall_start, L.edges.begin() + L.wall_start + L.n_wall, before);
}

HostMesh load_mesh(const std::string &input_dat, const std::string &directory, int duplicate, bool legacy_ordering)
{
    auto join = [&](const std::string &p) { return directory.empty() ? p : directory + "/" + p; };
    const InputDat in = parse_input_dat(join(input_dat));
    HostMesh M;
    M.size = in.size;
    M.mesh_variant = in.mesh_variant;
    M.mesh_name = in.mesh_name;
    M.level_files = in.level_files;
    M.map_files = in.map_files;
    M.levels.reserve(static_cast<size_t>(in.num_levels));
    // Every file is parsed by a task of its own (a level file, its coordinates, the map to the next level: none needs
    // another); the results are taken, and errors raised, in the order the serial reader met the files.
    const size_t nl = static_cast<size_t>(in.num_levels > 0 ? in.num_levels : 0);
    std::vector<std::future<LevelParse>> level_tasks(nl);
    std::vector<std::future<std::vector<double>>> coord_tasks(nl);
    std::vector<std::future<std::vector<int64_t>>> map_tasks(nl);
    std::vector<char> want_coords(nl, 0);
    for (size_t l = 0; l < nl; l++) {
        // Deliberate deviation, documented in DESIGN.md: the reference reads .coords only
        // when levels > 1 and then feeds uninitialised coordinates to adjust_ewt on
        // single-level m6wing/la_cascade/rotor37 runs.  We also read the
        // file for those when it exists.
        bool want = in.num_levels > 1;
        if (!want && in.mesh_variant != MGCFD_MESH_FVCORR) {
            if (FILE *c = std::fopen((join(in.level_files[l]) + ".coords").c_str(), "rb")) {
                std::fclose(c);
                want = true;
            }
        }
        want_coords[l] = want ? 1 : 0;
        level_tasks[l] = std::async(std::launch::async, parse_level_file, join(in.level_files[l]), in.mesh_variant);
        if (want) coord_tasks[l] = std::async(std::launch::async, parse_coords_file, join(in.level_files[l]) + ".coords");
        if (static_cast<int>(l) < in.num_levels - 1) map_tasks[l] = std::async(std::launch::async, read_mg_map, join(in.map_files[l]));
    }
    std::exception_ptr first_error;
    for (size_t l = 0; l < nl; l++) {
        // (after a failure the remaining tasks are still waited for: their futures block in their destructors anyway)
        try {
            LevelParse p = level_tasks[l].get();
            HostLevel L = finish_level(std::move(p), want_coords[l] ? &coord_tasks[l] : nullptr, join(in.level_files[l]));
            if (legacy_ordering) sort_edges_legacy(L);
            if (static_cast<int>(l) < in.num_levels - 1) L.mg_map = map_tasks[l].get();
            if (!first_error) M.levels.push_back(std::move(L));
        } catch (...) {
            if (!first_error) first_error = std::current_exception();
            if (coord_tasks[l].valid()) coord_tasks[l].wait();
            if (map_tasks[l].valid()) map_tasks[l].wait();
        }
    }
    if (first_error) std::rethrow_exception(first_error);
    if (duplicate > 1) {
        M.size *= duplicate;
        std::vector<int64_t> above(M.levels.size(), 0);
        for (size_t l = 0; l + 1 < M.levels.size(); l++) above[l] = M.levels[l + 1].nel;
        for (size_t l = 0; l < M.levels.size(); l++) duplicate_level(M.levels[l], duplicate, above[l]);
    }
    return M;
}

void write_array(const std::string &path, const double *data, int64_t nel, int ncols)
{
    FILE *f = std::fopen(path.c_str(), "w");
    if (!f) fail("ERROR: Failed to open file for writing: '" + path + "'");
    for (int64_t i = 0; i < nel; i++) {
        for (int c = 0; c < ncols; c++)
            std::fprintf(f, c + 1 < ncols ? "%.17e " : "%.17e\n", data[i * ncols + c]);
    }
    std::fclose(f);
}

std::vector<double> read_array(const std::string &path, int64_t nel, int ncols)
{
    TokenFile f(path);
    std::vector<double> out(static_cast<size_t>(nel * ncols));
    for (auto &v : out) v = f.next_double();
    return out;
}

int64_t identify_differences(const double *test_values, const double *master_values, int64_t nel, int mesh_variant)
{
    // Tolerance rule of the reference's -v check: per value max(|master|*1e-8, floor),
    // floor 3e-19, relaxed to 1e-15 for fvcorr (src/Kernels/validation.cpp:157-179).
    const double floor_abs = mesh_variant == MGCFD_MESH_FVCORR ? 1.0e-15 : 3.0e-19;
    for (int64_t k = 0; k < nel * MGCFD_NVAR; k++) {
        double tol = std::fabs(master_values[k] * 10.0e-9);
        if (tol < floor_abs) tol = floor_abs;
        if (std::fabs(test_values[k] - master_values[k]) > tol) return k;
    }
    return -1;
}

} // namespace mgcfd
