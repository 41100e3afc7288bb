// o2v_cli.cpp -- command line front end over the public C API (include/obj2voxel.h only), flag-compatible with the
// subset of the reference CLI that does not depend on its absent argument parser:
//   obj2voxel-amd INPUT_FILE OUTPUT_FILE -r RES [-s max|blend] [-p PERM] [-u] [-j THREADS] [-i FMT] [-o FMT]
//                 [-t TEXTURE] [-v] [-h]
// Reference: src/main.cpp:115-202 (mainImpl: the API call sequence), :224-262 (axis permutation), :291-312 (flags).
// -j is accepted for compatibility: worker threads are started and joined exactly like the reference does, but the
// MI355X path does not dispatch work to them.
#include "../../include/obj2voxel.h"

#include <cctype>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

void usage()
{
    std::puts("Usage: obj2voxel-amd INPUT_FILE OUTPUT_FILE -r RESOLUTION [options]\n"
              "  -r, --res N         maximum voxel grid resolution on any axis (required)\n"
              "  -s, --strat S       max | blend (default: max)\n"
              "  -p, --perm XYZ      permutation of the model's axes, capital letter flips (default: xyz)\n"
              "  -u, --super         2x supersampling\n"
              "  -j, --threads N     worker threads to start (kept for compatibility; the GPU path ignores them)\n"
              "  -i FMT / -o FMT     explicit input (obj|stl) / output (ply|vl32|xyzrgb) format\n"
              "  -t TEXTURE          fallback texture (png) for faces with uv coordinates but no material\n"
              "  -v, --verbose       debug logging\n"
              "  -V, --version       version and build information\n"
              "  --80                accepted for compatibility (the reference prints its help 80 columns wide)\n"
              "  -h, --help          this text");
}

// The -p argument: three letters, one per output axis, naming the model axis it takes its values from; a capital letter
// flips it ("xZy": x = x, y = -z, z = y).  Same rule as the reference's -p (src/main.cpp:224-262): every model axis must
// be used exactly once.  Row i of the result is the signed unit vector of the axis letter i selects.
bool parse_permutation(const std::string &str, int out[9])
{
    static const std::string letters = "xyzXYZ";
    if (str.size() != 3) return false;
    unsigned used = 0;
    std::fill(out, out + 9, 0);
    for (size_t row = 0; row < 3; ++row) {
        const size_t at = letters.find(str[row]);
        if (at == std::string::npos) return false;
        out[row * 3 + at % 3] = at < 3 ? 1 : -1;
        used |= 1u << (at % 3);
    }
    return used == 7u;
}

}  // namespace

int main(int argc, char **argv)
{
    const auto start = std::chrono::steady_clock::now();
    std::vector<std::string> positional;
    std::string in_format, out_format, texture_file, perm = "xyz", strat = "max";
    unsigned resolution = 0, threads = 0;
    bool supersample = false, verbose = false, have_res = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&](const char *name) -> const char * {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "missing value for %s\n", name);
                std::exit(1);
            }
            return argv[++i];
        };
        if (a == "-h" || a == "--help") {
            usage();
            return 0;
        }
        else if (a == "-r" || a == "--res") {
            resolution = (unsigned) std::strtoul(value("-r"), nullptr, 10);
            have_res = true;
        }
        else if (a == "-s" || a == "--strat") strat = value("-s");
        else if (a == "-p" || a == "--perm") perm = value("-p");
        else if (a == "-u" || a == "--super") supersample = true;
        else if (a == "-j" || a == "--threads") threads = (unsigned) std::strtoul(value("-j"), nullptr, 10);
        else if (a == "-i") in_format = value("-i");
        else if (a == "-o") out_format = value("-o");
        else if (a == "-t") texture_file = value("-t");
        else if (a == "-v" || a == "--verbose") verbose = true;
        else if (a == "-V" || a == "--version") {
            // the reference prints its header, version and compiler builtins (src/main.cpp:320-340); here: which device path
            std::puts("===== obj2voxel-amd =====");
            std::puts("Version:  1.3.5-dev (C API of obj2voxel 1.3.5-dev)");
            std::puts("Device:   HIP, gfx950 (MI355X); no CPU voxelization path");
            return 0;
        }
        else if (a == "--80") {
        }
        else if (!a.empty() && a[0] == '-') {
            std::fprintf(stderr, "unknown option %s\n", a.c_str());
            usage();
            return 1;
        }
        else positional.push_back(a);
    }
    if (positional.size() != 2 || !have_res || resolution == 0) {
        usage();
        return 1;
    }
    obj2voxel_enum_t strategy;
    if (strat == "max") strategy = OBJ2VOXEL_MAX_STRATEGY;
    else if (strat == "blend") strategy = OBJ2VOXEL_BLEND_STRATEGY;
    else {
        std::fprintf(stderr, "strategy must be max or blend\n");
        return 1;
    }
    int unit_transform[9];
    if (!parse_permutation(perm, unit_transform)) {
        std::fprintf(stderr, "invalid permutation \"%s\"\n", perm.c_str());
        return 1;
    }
    if (verbose) obj2voxel_set_log_level(OBJ2VOXEL_LOG_LEVEL_DEBUG);

    // the API call sequence of the reference's mainImpl (src/main.cpp:147-200)
    obj2voxel_instance *instance = obj2voxel_alloc();
    std::vector<std::thread> workers;
    for (unsigned i = 0; i < threads; ++i) workers.emplace_back(&obj2voxel_run_worker, instance);
    obj2voxel_set_parallel(instance, threads != 0);
    obj2voxel_set_input_file(instance, positional[0].c_str(), in_format.empty() ? nullptr : in_format.c_str());
    obj2voxel_set_output_file(instance, positional[1].c_str(), out_format.empty() ? nullptr : out_format.c_str());
    obj2voxel_texture *texture = nullptr;
    if (!texture_file.empty()) {
        texture = obj2voxel_texture_alloc();
        if (obj2voxel_texture_load_from_file(texture, texture_file.c_str(), nullptr)) obj2voxel_set_texture(instance, texture);
        else std::fprintf(stderr, "continuing without fallback texture because it could not be loaded\n");
    }
    obj2voxel_set_unit_transform(instance, unit_transform);
    obj2voxel_set_resolution(instance, resolution);
    obj2voxel_set_supersampling(instance, supersample ? 2 : 1);
    obj2voxel_set_color_strategy(instance, strategy);
    const obj2voxel_error_t result = obj2voxel_voxelize(instance);
    obj2voxel_stop_workers(instance);
    for (std::thread &w : workers) w.join();
    if (texture) obj2voxel_texture_free(texture);
    obj2voxel_free(instance);

    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    std::printf("%s (%.3f s)\n", result == OBJ2VOXEL_ERR_OK ? "Done!" : "Failed", secs);
    // The output file is written and closed (obj2voxel_voxelize / obj2voxel_free); what is left is the teardown of the HIP
    // runtime behind the library's cached device session (static destructors, unmapping the dense grids): tens of
    // milliseconds of a run that takes a few hundred, for a process that is about to vanish.  Streams flushed, then out.
    std::fflush(nullptr);
    std::_Exit(result);
}
