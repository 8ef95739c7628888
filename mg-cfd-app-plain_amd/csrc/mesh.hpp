// mesh.hpp — host-side multigrid input: the reference's file formats parsed into
// std::vectors (reference readers: src/Base/io.cpp:14-199, src/Base/io_enhanced.cpp:89-201,
// :407-579, :629-650).  No GPU dependency; part of libmgcfd_hip.so's host half.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "mgcfd.h"

namespace mgcfd {

struct HostLevel {
    int64_t nel = 0;
    int64_t n_internal = 0, n_boundary = 0, n_wall = 0;
    int64_t internal_start = 0, boundary_start = 0, wall_start = 0;
    std::vector<double> volumes;        // [nel]
    std::vector<double> coords;         // [nel*3] (zeros when the .coords file was not read)
    bool have_coords = false;
    std::vector<mgcfd_edge> edges;      // [number_of_edges] internal | boundary | wall (-5 padded)
    std::vector<int64_t> mg_map;        // fine -> coarse (next level); empty on the last level

    mgcfd_level_desc desc() const;
};

struct HostMesh {
    int size = 0;                       // input.dat "size" (x duplicate count)
    int mesh_variant = -1;
    std::string mesh_name;
    std::vector<std::string> level_files, map_files;
    std::vector<HostLevel> levels;
};

struct InputDat {
    int size = 0, num_levels = 0, mesh_variant = -1;
    std::string mesh_name;
    std::vector<std::string> level_files, map_files;
};

// All of these throw std::runtime_error with a message on failure.
InputDat parse_input_dat(const std::string &path);
HostLevel read_mesh_level(const std::string &path, int mesh_variant, bool read_coords);
std::vector<int64_t> read_mg_map(const std::string &path);
void duplicate_level(HostLevel &lvl, int copies, int64_t nel_above);
// legacy_ordering: sort every edge class by (a, b, x, y, z) as the reference's -DLEGACY_ORDERING build
// does at the end of read_grid (src/Base/io.cpp:183-193, comparator src/Base/common.h:145-157).
void sort_edges_legacy(HostLevel &lvl);
HostMesh load_mesh(const std::string &input_dat, const std::string &directory, int duplicate, bool legacy_ordering = false);

void write_array(const std::string &path, const double *data, int64_t nel, int ncols);
std::vector<double> read_array(const std::string &path, int64_t nel, int ncols);
int64_t identify_differences(const double *test_values, const double *master_values, int64_t nel,
                             int mesh_variant);

const char *mesh_variant_name(int variant);

} // namespace mgcfd
