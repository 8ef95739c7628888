// Host-only: the file reader (mesh.cpp: mapped files, std::from_chars, a task per file) under a sanitizer, on every golden input and
// on truncated copies of them (a file that ends inside a token, inside a record, after the header):
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -Iinclude -Img-cfd-app-plain_amd/csrc tools/reader_sanitize.cpp mg-cfd-app-plain_amd/csrc/mesh.cpp -lpthread -o /tmp/reader_asan
//   /tmp/reader_asan tests/golden/m6_3lvl/input [more directories]      (-fsanitize=thread likewise)
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <string>
#include "mesh.hpp"
namespace fs = std::filesystem;
static int load(const std::string &dir, const char *what)
{
    try {
        const mgcfd::HostMesh m = mgcfd::load_mesh("input.dat", dir, 1);
        std::printf("%s: %s: %zu levels, level 0: %ld nodes\n", dir.c_str(), what, m.levels.size(), m.levels.empty() ? 0L : (long)m.levels[0].nel);
        return 0;
    } catch (const std::exception &e) {
        std::printf("%s: %s: refused: %.90s\n", dir.c_str(), what, e.what());
        return 1;
    }
}
int main(int argc, char **argv)
{
    for (int a = 1; a < argc; a++) {
        const std::string dir = argv[a];
        load(dir, "as it is");
        // truncated copies: every file cut at a few lengths (the mapping then ends inside a number, a record, the header)
        const fs::path tmp = fs::temp_directory_path() / ("reader_sanitize_" + std::to_string(a));
        for (const double cut : {0.999, 0.5, 0.02, 0.0}) {
            fs::remove_all(tmp); fs::create_directories(tmp);
            for (const auto &f : fs::directory_iterator(dir)) {
                if (!f.is_regular_file()) continue;
                std::ifstream in(f.path(), std::ios::binary);
                std::string bytes((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
                const bool is_input = f.path().filename() == "input.dat";
                std::ofstream(tmp / f.path().filename(), std::ios::binary).write(bytes.data(), is_input ? bytes.size() : static_cast<std::streamsize>(bytes.size() * cut));
            }
            load(tmp.string(), ("every mesh file cut to " + std::to_string(cut) + " of its length").c_str());
        }
        fs::remove_all(tmp);
    }
    return 0;
}
