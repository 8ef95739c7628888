// euler3d_gpu_double — drop-in replacement for the reference's euler3d_cpu_double binary
// (src/euler3d_cpu_double.cpp) on one MI355X, written purely against the C ABI of
// include/mgcfd.h.  Same command line (src/Base/config.cpp:32-47), same input.dat / mesh /
// .coords / MG-map inputs, same stdout progress lines, same `variables` dump, and the same
// Times.csv / LoopNumIters.csv schema (src/Monitoring/timer.cpp:106-195,
// src/Monitoring/loop_stats.cpp:83-171, identification columns src/Base/io_enhanced.cpp:858-1016).
#include <getopt.h>
#include <sched.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "mgcfd.h"
#include "multi_gpu.hpp"

namespace {

struct Config {                       // the reference's `config` (src/Base/config.h:27-47)
    std::string config_filepath, input_file, input_file_directory, papi_config_file, output_file_prefix;
    int mesh_duplicate_count = 1;
    int num_cycles = 25;              // src/Base/config.cpp:63
    bool validate_result = false;
    bool output_variables = false, output_old_variables = false, output_step_factors = false,
         output_edge_fluxes = false, output_fluxes = false, output_volumes = false;
    // extensions (not in the reference)
    bool timers = true;               // --no-timers: fused fast path (one launch per Runge-Kutta stage); Times.csv holds only Total
    bool loop_timers = false;         // --loop-timers: EVERY loop its own launch between two events, as the reference's -DTIME build brackets
                                      // them (src/Monitoring/timer.cpp:58-195); default: fused stages, the per-loop times ATTRIBUTED
                                      // from every 32nd sweep / transfer of a level, which runs per loop under events (MGCFD_OPT_TIMING = 4)
    bool fast_math = false;           // --fast: allow FMA contraction (MGCFD_OPT_EXACT = 0)
    bool indirect_rw = true;          // the reference runs the probe every RK stage; --no-indirect-rw skips it
    bool legacy_ordering = false;     // --legacy-ordering: the reference's -DLEGACY_ORDERING edge sort (a compile-time flag there)
    int device = 0;
    int gpus = 1;                     // --gpus N: a single-level input partitioned over N GPUs, a multigrid input one level per GPU
    bool gpus_share_device = false;   // --gpus-share-device: all N ranks on --device (rehearsal on a one-GPU box)
    bool gpus_partition = false;      // --gpus-partition: split every level of a multigrid input over the N GPUs (the default when N > levels)
};

std::string trim(const std::string &s)
{
    size_t b = s.find_first_not_of(" \t\r\n");
    if (b == std::string::npos) return "";
    return s.substr(b, s.find_last_not_of(" \t\r\n") - b + 1);
}

void set_param(Config &c, const std::string &key, const std::string &value)
{
    // src/Base/config.cpp:81-157
    if (key == "config_filepath") c.config_filepath = value;
    else if (key == "input_file") c.input_file = value;
    else if (key == "input_file_directory") c.input_file_directory = value;
    else if (key == "papi_config_file") c.papi_config_file = value;
    else if (key == "output_file_prefix") c.output_file_prefix = value;
    else if (key == "mesh_duplicate_count") c.mesh_duplicate_count = std::atoi(value.c_str());
    else if (key == "cycles") c.num_cycles = std::atoi(value.c_str());
    else if (key == "omp_num_threads") { /* no OpenMP here */ }
    else if (key == "output_variables") { if (value == "Y") c.output_variables = true; }
    else if (key == "output_old_variables") { if (value == "Y") c.output_old_variables = true; }
    else if (key == "output_step_factors") { if (value == "Y") c.output_step_factors = true; }
    else if (key == "output_edge_fluxes") { if (value == "Y") c.output_edge_fluxes = true; }
    else if (key == "output_fluxes") { if (value == "Y") c.output_fluxes = true; }
    else if (key == "output_volumes") { if (value == "Y") c.output_volumes = true; }
    else std::printf("WARNING: Unknown key '%s' encountered during parsing of config file.\n", key.c_str());
}

void read_config(Config &c)
{
    // src/Base/config.cpp:159-217
    if (access(c.config_filepath.c_str(), F_OK) == -1) {
        std::fprintf(stderr, "ERROR: \"%s\" does not exist.\n", c.config_filepath.c_str());
        return;
    }
    std::ifstream file(c.config_filepath);
    std::string line;
    while (std::getline(file, line)) {
        if (!line.empty() && line[0] == '#') continue;
        size_t eq = line.find('=');
        if (eq == std::string::npos || eq + 1 >= line.size()) continue;
        set_param(c, trim(line.substr(0, eq)), trim(line.substr(eq + 1)));
    }
    std::string dir;
    size_t slash = c.config_filepath.rfind('/');
    if (slash != std::string::npos) dir = c.config_filepath.substr(0, slash);
    if ((c.input_file_directory.empty() || c.input_file_directory[0] != '/') && !dir.empty()) {
        if (c.input_file_directory == "./") c.input_file_directory = dir;
        else c.input_file_directory = dir + "/" + c.input_file_directory;
    }
}

void print_help()
{
    std::fprintf(stderr,
        "MG-CFD (MI355X) instructions\n\n"
        "Usage: euler3d_gpu_double [OPTIONS] \n\n"
        "  -h, --help                       Print help\n\n"
        "CRITICAL ARGUMENTS\n  One of these must be set:\n"
        "  -i, --input-file=FILEPATH        Multigrid input grid (.dat file)\n"
        "  -c, --config-filepath=FILEPATH   Config file\n\n"
        "OPTIONAL ARGUMENTS\n"
        "  -d, --input-directory=DIRPATH    Directory path to input files\n"
        "  -o, --output-file-prefix=STRING  String to prepend to output filenames\n"
        "  -p, --papi-config-file=FILEPATH  Accepted and ignored (CPU performance counters)\n\n"
        "  -g, --num-cycles=INT             Number of multigrid V-cycles\n"
        "  -m, --mesh-duplicate-count=INT   Number of times to duplicate mesh\n"
        "  -v, --validate-result            Check final state against pre-calculated solution\n\n"
        "DEBUGGING ARGUMENTS\n"
        "  --output-variables               Write Euler equation variable values to file\n"
        "  --output-fluxes                  Write flux accumulations to file\n"
        "  --output-step-factors            Write step factors to file\n\n"
        "GPU ARGUMENTS (extensions)\n"
        "  --device=INT                     GPU to run on (default 0); with --gpus the first of the N devices\n"
        "  --gpus=INT                       Run on N GPUs of this node: a single-level input is partitioned over them\n"
        "                                   (halo messages after every Runge-Kutta stage), a multigrid input runs one\n"
        "                                   level per GPU; fused path (as --no-timers)\n"
        "  --gpus-share-device              With --gpus: every rank on the one device (functional rehearsal)\n"
        "  --gpus-partition                 With --gpus on a multigrid input: split EVERY level over the N GPUs and run the\n"
        "                                   whole V-cycle inside the library (the default when N exceeds the number of levels)\n"
        "  --no-timers                      One fused launch per Runge-Kutta stage; no per-loop times\n"
        "  --loop-timers                    Every loop its own launch between two events (2.7x slower cycles); default: fused\n"
        "                                   stages, per-loop times attributed from every 32nd sweep, which runs per loop\n"
        "  --no-indirect-rw                 Skip the indirect_rw bandwidth probe each RK stage\n"
        "  --fast                           Fast mode: FMA contraction and order-free flux accumulation (results within\n"
        "                                   1e-12 relative of the reference's per sweep, not reproducible bit for bit from run to run)\n"
        "  --legacy-ordering                Sort edges by (a,b,x,y,z) like the reference built with -DLEGACY_ORDERING\n");
}

bool parse_arguments(int argc, char **argv, Config &c)
{
    // src/Base/config.cpp:32-47,219-259.  (The reference stores the three --output-* flags
    // through bool-to-int* casts, so each one also clobbers the bools after it; here every
    // flag sets exactly its own field.)
    static const option long_opts[] = {
        {"help", no_argument, nullptr, 'h'},
        {"config-filepath", required_argument, nullptr, 'c'},
        {"input-file", required_argument, nullptr, 'i'},
        {"input-directory", required_argument, nullptr, 'd'},
        {"papi_config_file", required_argument, nullptr, 'p'},
        {"output-file-prefix", required_argument, nullptr, 'o'},
        {"mesh-duplicate-count", required_argument, nullptr, 'm'},
        {"num-cycles", required_argument, nullptr, 'g'},
        {"validate-result", no_argument, nullptr, 'v'},
        {"output-variables", no_argument, nullptr, 1001},
        {"output-fluxes", no_argument, nullptr, 1002},
        {"output-step-factors", no_argument, nullptr, 1003},
        {"device", required_argument, nullptr, 1004},
        {"no-timers", no_argument, nullptr, 1005},
        {"no-indirect-rw", no_argument, nullptr, 1006},
        {"fast", no_argument, nullptr, 1007},
        {"legacy-ordering", no_argument, nullptr, 1008},
        {"gpus", required_argument, nullptr, 1009},
        {"gpus-share-device", no_argument, nullptr, 1010},
        {"loop-timers", no_argument, nullptr, 1011},
        {"gpus-partition", no_argument, nullptr, 1012},
        {nullptr, 0, nullptr, 0}};
    int optc;
    while ((optc = getopt_long(argc, argv, "hc:i:d:p:o:m:g:v", long_opts, nullptr)) != -1) {
        switch (optc) {
            case 'h': print_help(); return false;
            case 'i': c.input_file = optarg; break;
            case 'c': c.config_filepath = optarg; read_config(c); break;
            case 'd': c.input_file_directory = optarg; break;
            case 'p': c.papi_config_file = optarg; break;
            case 'o': c.output_file_prefix = optarg; break;
            case 'm': c.mesh_duplicate_count = std::atoi(optarg); break;
            case 'g': c.num_cycles = std::atoi(optarg); break;
            case 'v': c.validate_result = true; break;
            case 1001: c.output_variables = true; break;
            case 1002: c.output_fluxes = true; break;
            case 1003: c.output_step_factors = true; break;
            case 1004: c.device = std::atoi(optarg); break;
            case 1005: c.timers = false; break;
            case 1006: c.indirect_rw = false; break;
            case 1007: c.fast_math = true; break;
            case 1008: c.legacy_ordering = true; break;
            case 1009: c.gpus = std::atoi(optarg); break;
            case 1010: c.gpus_share_device = true; break;
            case 1011: c.loop_timers = true; break;
            case 1012: c.gpus_partition = true; break;
            default: std::printf("Unknown command line parameter '%c'\n", optc);
        }
    }
    return true;
}

// src/Base/io_enhanced.cpp:26-74
std::string filename_suffix(const Config &c, int level)
{
    std::string s = "size=" + std::to_string(c.mesh_duplicate_count) + "x.cycles=" + std::to_string(c.num_cycles);
    if (level >= 0) s += ".level=" + std::to_string(level);
    return s;
}
std::string output_filepath(const Config &c, const std::string &name, int level)
{
    std::string p = c.output_file_prefix;
    if (!p.empty() && p.back() != '/') p += ".";
    return p + name + "." + filename_suffix(c, level);
}
std::string solution_filepath(const Config &c, const std::string &name, int level)
{
    std::string p = c.input_file_directory;
    if (!p.empty() && p.back() != '/') p += "/";
    return p + "solution." + name + "." + filename_suffix(c, level);
}
std::string csv_filepath(const Config &c, const char *name)
{
    std::string p = c.output_file_prefix;
    if (!p.empty() && p.back() != '/') p += ".";
    return p + name;
}

const char *mesh_name(int v)
{
    switch (v) {
        case MGCFD_MESH_LA_CASCADE: return "la_cascade";
        case MGCFD_MESH_ROTOR_37: return "rotor37";
        case MGCFD_MESH_FVCORR: return "fvcorr";
        case MGCFD_MESH_M6_WING: return "m6wing";
        default: return "unknown";
    }
}

#define STR2(x) #x
#define STR(x) STR2(x)

// The 16 identification columns of src/Base/io_enhanced.cpp:858-1016, re-read for a GPU build:
// CC = hipcc's clang, Instruction set = gfx950, Num threads = number of GPUs, CPU = device name.
void csv_identification(const Config &c, int size, int mesh_variant, const std::string &device_name,
                        std::string &header, std::string &line, int num_gpus = 1)
{
    std::ostringstream h, d;
    h << "Size,";                  d << size << ",";
    h << "Mesh,";                  d << mesh_name(mesh_variant) << ",";
    h << "MG cycles,";             d << c.num_cycles << ",";
    h << "Flux variant,";          d << "Normal,";
    // (the reference leaves the column empty for a plain build; here it says how the per-loop times were obtained)
    h << "Flux options,";          d << (c.timers && !c.loop_timers && num_gpus == 1 ? "fused stages; loop times attributed from every 32nd sweep run per loop" : "") << ",";
    h << "CC,";                    d << "hipcc,";
    h << "CC version,";            d << STR(__clang_major__) "." STR(__clang_minor__) "." STR(__clang_patchlevel__) ",";
    h << "Opt level,";             d << "3,";
    h << "Instruction set,";       d << "gfx950,";
    h << "SIMD,";                  d << "N,";
    h << "SIMD len,";              d << "1,";
    h << "OpenMP,";                d << "Off,";
    h << "Num threads,";           d << num_gpus << ",";
    h << "Permit scatter OpenMP,"; d << "N,";
    h << "Flux fission,";          d << "N,";
    h << "CPU,";                   d << device_name << ",";
    header = h.str();
    line = d.str();
}

void write_csv(const std::string &path, const std::string &ident_header, const std::string &ident_line, int levels,
               const std::vector<std::vector<std::string>> &cells, bool with_total, double total)
{
    std::remove(path.c_str());
    std::ofstream out(path);
    static const char *cols[MGCFD_NUM_LOOPS] = {"flux", "update", "compute_step", "time_step", "restrict", "prolong", "indirect_rw"};
    out << ident_header << "ThreadNum,CpuId,";
    for (int l = 0; l < levels; l++)
        for (int k = 0; k < MGCFD_NUM_LOOPS; k++) out << cols[k] << l << ",";
    if (with_total) out << "Total,";
    out << std::endl;
    out << ident_line << 0 << "," << sched_getcpu() << ",";
    for (int l = 0; l < levels; l++)
        for (int k = 0; k < MGCFD_NUM_LOOPS; k++) out << cells[l][k] << ",";
    if (with_total) out << total << ",";
    out << std::endl;
}

int fail(const char *what)
{
    std::fprintf(stderr, "ERROR: %s: %s\n", what, mgcfd_last_error());
    return EXIT_FAILURE;
}

// What the reference does between its cycle loop and its CSV files (src/euler3d_cpu_double.cpp:704-772), for one GPU or
// several: -v / validate_result = Y (NaN check of every level, then level 0 against the solution file with
// identify_differences' tolerance) and the level-0 dumps.  get(which, ncols, out) reads a level-0 array of the WHOLE mesh in
// original numbering; nan_check(level, &cell) is check_for_invalid_variables on that level.  Returns 0 or EXIT_FAILURE.
template <typename Get, typename NanCheck>
int validate_and_dump(const Config &conf, int levels, int mesh_variant, int64_t nel0, Get &&get, NanCheck &&nan_check)
{
    std::printf("\n");
    if (conf.validate_result) {
        std::printf("Beginning validation of variables[]\n");
        for (int l = 0; l < levels; l++) {
            int64_t bad = -1;
            if (nan_check(l, &bad) != MGCFD_OK) {
                std::printf("\nERROR: NaN detected!\nCell %ld\n", (long)bad);
                return EXIT_FAILURE;
            }
        }
        std::printf("  NaN check passed\n");
        bool passed = true;
        const std::string sol = solution_filepath(conf, "variables", 0);
        std::ifstream file(sol);
        if (!file.is_open()) {
            std::printf("  could not open variables solution file:\n    %s\n  aborting validation\n", sol.c_str());
            passed = false;
        } else {
            std::vector<double> variables(static_cast<size_t>(nel0) * MGCFD_NVAR), master(variables.size());
            for (auto &v : master) file >> v;
            if (get(MGCFD_ARR_VARIABLES, MGCFD_NVAR, variables.data()) != MGCFD_OK) return fail("reading back variables");
            std::printf("  scanning variables[] on level 0 for errors\n");
            int64_t first_bad = -1;
            if (mgcfd_identify_differences(variables.data(), master.data(), nel0, mesh_variant, &first_bad) != MGCFD_OK) {
                std::printf("ERROR: Unacceptable error detected at (i=%ld, v=%d)\n", (long)(first_bad / MGCFD_NVAR), int(first_bad % MGCFD_NVAR));
                std::printf("       - incorrect value = %.23f\n", variables[static_cast<size_t>(first_bad)]);
                std::printf("       - correct value =   %.23f\n", master[static_cast<size_t>(first_bad)]);
                return EXIT_FAILURE;
            }
        }
        if (passed) std::printf("PASS: variables[] validated successfully\n");
        std::printf("\n");
    }
    // ---- dumps, level 0 only (src/euler3d_cpu_double.cpp:752-772) ----
    auto dump = [&](int which, const char *name, int ncols) -> int {
        std::vector<double> a(static_cast<size_t>(nel0) * ncols);
        if (get(which, ncols, a.data()) != MGCFD_OK) return fail("reading back an array");
        const std::string path = output_filepath(conf, name, 0);
        if (which == MGCFD_ARR_VARIABLES) std::printf("Dumping variables[] to file: %s\n", path.c_str());
        if (mgcfd_write_array(path.c_str(), a.data(), nel0, ncols) != MGCFD_OK) return fail("writing a dump");
        return 0;
    };
    if (conf.output_variables && dump(MGCFD_ARR_VARIABLES, "variables", MGCFD_NVAR)) return EXIT_FAILURE;
    if (conf.output_step_factors && dump(MGCFD_ARR_STEP_FACTORS, "step_factors", 1)) return EXIT_FAILURE;
    if (conf.output_fluxes && dump(MGCFD_ARR_FLUXES, "fluxes", MGCFD_NVAR)) return EXIT_FAILURE;
    if (conf.output_volumes && dump(MGCFD_ARR_VOLUMES, "volumes", 1)) return EXIT_FAILURE;
    return 0;
}

// --gpus N (multi_gpu.cpp): the same outputs as the one-GPU run from N ranks of this process
int run_on_several_gpus(const Config &conf, mgcfd_mesh *mesh, int levels, int mesh_variant, int problem_size)
{
    try {
        multi_gpu::Options o;
        o.gpus = conf.gpus; o.first_device = conf.device; o.share_device = conf.gpus_share_device; o.fast_math = conf.fast_math;
        o.partition_levels = conf.gpus_partition;
        const auto tb = std::chrono::steady_clock::now();
        multi_gpu::Run run(mesh, o);
        std::fprintf(stderr, "[euler3d_gpu_double] %d ranks (%s), set up in %.2f s\n", run.ranks(),
                     run.partitioned_hierarchy() ? "every level partitioned, level 0 by recursive coordinate bisection; the V-cycle inside the library" :
                     run.partitioned() ? "level 0 partitioned by recursive coordinate bisection" : "one multigrid level per GPU",
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - tb).count());
        std::vector<double> rms(static_cast<size_t>(conf.num_cycles > 0 ? conf.num_cycles : 0));
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = run.run_cycles(conf.num_cycles, rms.data());
        const double total_compute_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (int i = 0; i < conf.num_cycles; i++)
            std::printf(levels <= 1 ? "\nCycle %d / %d (RMS = %.3e)" : "\nMG cycle %d / %d (RMS = %.3e)", i + 1, conf.num_cycles, rms[static_cast<size_t>(i)]);
        std::printf("\n");
        if (rc == MGCFD_ERR_NAN || rc == MGCFD_ERR_NEG_DENSITY || rc == MGCFD_ERR_NEG_ENERGY) {
            std::printf(rc == MGCFD_ERR_NAN ? "\nERROR: NaN detected!\n" : rc == MGCFD_ERR_NEG_DENSITY ? "\nERROR: Negative density detected!\n" : "\nERROR: Negative density.energy detected!\n");
            return EXIT_FAILURE;
        }
        if (rc != MGCFD_OK) return fail("running the cycles");
        std::printf("Total runtime = %g\n", total_compute_time);
        mgcfd_level_desc d0;
        mgcfd_mesh_level(mesh, 0, &d0);
        // -v and every dump as on one GPU (level 0 gathered from the ranks that own its nodes)
        if (validate_and_dump(conf, levels, mesh_variant, d0.nel,
                              [&](int which, int ncols, double *out) { run.get_level0(which, ncols, out); return MGCFD_OK; },
                              [&](int level, int64_t *bad) { return run.check_invalid(level, bad); }))
            return EXIT_FAILURE;
        std::string device_name = "unknown GPU";
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, conf.device) == hipSuccess) device_name = prop.name;
        std::string ih, il;
        csv_identification(conf, problem_size, mesh_variant, device_name, ih, il, run.ranks());
        std::vector<std::vector<std::string>> times(static_cast<size_t>(levels)), iters(static_cast<size_t>(levels));
        for (int l = 0; l < levels; l++) {
            int64_t n[MGCFD_NUM_LOOPS];
            run.loop_iters(l, conf.num_cycles, n);
            for (int k = 0; k < MGCFD_NUM_LOOPS; k++) { times[static_cast<size_t>(l)].push_back("0"); iters[static_cast<size_t>(l)].push_back(std::to_string(n[k])); }
        }
        const std::string tpath = csv_filepath(conf, "Times.csv"), ipath = csv_filepath(conf, "LoopNumIters.csv");
        write_csv(tpath, ih, il, levels, times, true, total_compute_time);
        std::printf("Loop runtimes written to: %s\n", tpath.c_str());
        write_csv(ipath, ih, il, levels, iters, false, 0.0);
        std::printf("Loop stats written to: %s\n", ipath.c_str());
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "ERROR: %s\n", e.what());
        return EXIT_FAILURE;
    }
}

} // namespace

int main(int argc, char **argv)
{
    const double epoch_at_main = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    Config conf;
    if (!parse_arguments(argc, argv, conf)) return 1;
    if (conf.input_file.empty()) {
        std::printf("ERROR: input_file not set\n");
        return 1;
    }

    const auto t_start = std::chrono::steady_clock::now();
    if (conf.gpus <= 1) mgcfd_device_warm_up(conf.device);      // (the runtime comes up while the files are read)
    mgcfd_mesh *mesh = nullptr;
    if (mgcfd_mesh_load_ex(conf.input_file.c_str(), conf.input_file_directory.c_str(), conf.mesh_duplicate_count,
                           conf.legacy_ordering ? MGCFD_MESH_LEGACY_ORDERING : 0, &mesh) != MGCFD_OK)
        return fail("reading input");
    const int levels = mgcfd_mesh_num_levels(mesh);
    const int mesh_variant = mgcfd_mesh_variant(mesh);
    const int problem_size = mgcfd_mesh_size(mesh);

    const double t_read = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    if (conf.gpus > 1) return run_on_several_gpus(conf, mesh, levels, mesh_variant, problem_size);

    mgcfd_solver *solver = nullptr;
    const auto t_create = std::chrono::steady_clock::now();
    if (mgcfd_create_from_mesh(mesh, conf.device, &solver) != MGCFD_OK) return fail("creating the GPU solver");
    // where the wall time outside the reference's timed region goes (stderr: stdout stays the reference's, line for line)
    std::fprintf(stderr, "[euler3d_gpu_double] input files read in %.2f s, gather plans built and uploaded in %.2f s\n", t_read,
                 std::chrono::duration<double>(std::chrono::steady_clock::now() - t_create).count());
    const auto t_created = std::chrono::steady_clock::now();
    mgcfd_set_option(solver, MGCFD_OPT_EXACT, conf.fast_math ? 0 : 1);
    mgcfd_set_option(solver, MGCFD_OPT_TIMING, conf.timers ? (conf.loop_timers ? 1 : 4) : 0);
    mgcfd_set_option(solver, MGCFD_OPT_INDIRECT_RW, (conf.indirect_rw && conf.timers) ? 1 : 0);

    std::string device_name = "unknown GPU";
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, conf.device) == hipSuccess) device_name = prop.name;
    }

    // ---- compute (src/euler3d_cpu_double.cpp:368-698) ----
    std::vector<double> rms(static_cast<size_t>(conf.num_cycles > 0 ? conf.num_cycles : 0));
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = mgcfd_run_cycles(solver, conf.num_cycles, rms.data());
    const double total_compute_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const bool invalid = rc == MGCFD_ERR_NAN || rc == MGCFD_ERR_NEG_DENSITY || rc == MGCFD_ERR_NEG_ENERGY;
    int64_t bad_cell = -1;
    int bad_cycle = -1;
    if (invalid) mgcfd_invalid_state_location(solver, &bad_cell, &bad_cycle);
    // (the reference prints a cycle's line when the cycle starts and exits inside the failing time_step)
    const int printed = invalid && bad_cycle >= 0 ? bad_cycle + 1 : conf.num_cycles;
    for (int i = 0; i < printed; i++) {
        std::printf(levels <= 1 ? "\nCycle %d / %d" : "\nMG cycle %d / %d", i + 1, conf.num_cycles);
        if (!(invalid && i == bad_cycle)) std::printf(" (RMS = %.3e)", rms[static_cast<size_t>(i)]);
    }
    std::printf("\n");
    if (invalid) {
        // check_for_invalid_variables' messages (src/Kernels/validation.cpp:112-134); the cell's values at that
        // moment are not kept on the device
        std::printf(rc == MGCFD_ERR_NAN ? "\nERROR: NaN detected!" :
                    rc == MGCFD_ERR_NEG_DENSITY ? "\nERROR: Negative density detected!" : "\nERROR: Negative density.energy detected!");
        std::printf("\nCell %ld\n", static_cast<long>(bad_cell));
        return EXIT_FAILURE;
    }
    if (rc != MGCFD_OK) return fail("running the cycles");
    std::printf("Total runtime = %g\n", total_compute_time);

    const int64_t nel0 = mgcfd_level_nel(solver, 0);
    // ---- validate, dumps (src/euler3d_cpu_double.cpp:704-772) ----
    if (validate_and_dump(conf, levels, mesh_variant, nel0,
                          [&](int which, int ncols, double *out) { (void)ncols; return mgcfd_get_array(solver, 0, which, out); },
                          [&](int level, int64_t *bad) { return mgcfd_check_for_invalid_variables(solver, level, bad); }))
        return EXIT_FAILURE;

    // ---- performance data (src/euler3d_cpu_double.cpp:778-785) ----
    std::string ih, il;
    csv_identification(conf, problem_size, mesh_variant, device_name, ih, il);
    std::vector<std::vector<std::string>> times(static_cast<size_t>(levels)), iters(static_cast<size_t>(levels));
    for (int l = 0; l < levels; l++) {
        double t[MGCFD_NUM_LOOPS];
        int64_t n[MGCFD_NUM_LOOPS];
        mgcfd_get_loop_times(solver, l, t);
        mgcfd_get_loop_iters(solver, l, n);
        for (int k = 0; k < MGCFD_NUM_LOOPS; k++) {
            std::ostringstream a, b;
            a << t[k];
            b << n[k];
            times[static_cast<size_t>(l)].push_back(a.str());
            iters[static_cast<size_t>(l)].push_back(b.str());
        }
    }
    const std::string tpath = csv_filepath(conf, "Times.csv"), ipath = csv_filepath(conf, "LoopNumIters.csv");
    write_csv(tpath, ih, il, levels, times, true, total_compute_time);
    std::printf("Loop runtimes written to: %s\n", tpath.c_str());
    write_csv(ipath, ih, il, levels, iters, false, 0.0);
    std::printf("Loop stats written to: %s\n", ipath.c_str());

    const auto t_out = std::chrono::steady_clock::now();
    mgcfd_destroy(solver);
    mgcfd_mesh_free(mesh);
    if (std::getenv("MGCFD_PLAN_TIMING"))       // (the two epoch times let a caller see what the process spends before main() and after it)
        std::fprintf(stderr, "[euler3d_gpu_double] main() began at %.3f and returns at %.3f (seconds of the epoch)\n", epoch_at_main,
                     std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count());
    if (std::getenv("MGCFD_PLAN_TIMING"))
        std::fprintf(stderr, "[euler3d_gpu_double] since main() began: files read %.3f s, solver created %.3f s, cycles done %.3f s, outputs written %.3f s, freed %.3f s\n",
                     t_read, std::chrono::duration<double>(t_created - t_start).count(), std::chrono::duration<double>(t0 - t_start).count() + total_compute_time,
                     std::chrono::duration<double>(t_out - t_start).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
    return 0;
}
