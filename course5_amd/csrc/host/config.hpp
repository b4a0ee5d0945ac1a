// Render configuration and physical constants of the `course` command line.
// Values and defaults follow the reference (project/include/config.hpp:15-28,39-74 and the
// option table at project/src/main.cpp:21-33): they are part of the CLI contract.
#pragma once

#include <cstddef>
#include <limits>
#include <string>

struct render_config {
    std::string file;
    std::string destination;
    int threads = 1;
    std::size_t resolution_x = 1200;  // main.cpp:26 default
    std::size_t resolution_y = 900;   // main.cpp:27 default
    double angle_around_x = 0;        // units of pi (main.cpp:88-91)
    double angle_around_y = 0;
    double donor_angle = 0;
    double system_initial_angle_around_y = 0;
    double limit_alpha_value = 2.5;
    double acc_disk_solid_color = std::numeric_limits<double>::quiet_NaN();
    double roche_lobe_solid_color = std::numeric_limits<double>::quiet_NaN();
    // additions of this build (all optional)
    int device = 0;
    std::string devices;          // "0-7", "0,1", ...: several GPUs of one node (overrides --device)
    std::string exchange = "host";  // how rows rendered by several GPUs come together: host, rccl or p2p (plane.hpp)
    std::string split = "auto";   // several GPUs: rows (every frame split by rows) or frames (frame k -> GPU k mod N);
                                  // auto = frames for a sweep, rows for a single frame
    std::string row_layout = "auto";  // rows split: blocks (one contiguous cost-balanced block per GPU), tiles (cyclic 16-row
                                      // tiles) or auto = blocks for a sweep, tiles for a single frame
    bool bench_files = false;     // --bench writes every timed frame like a sweep does (end-to-end figure with files)
    bool auto_bounds = false;     // image domain = bounding box of the transformed objects (plane.cpp:278-288) instead of main.cpp:83's
    std::size_t bench = 0;        // > 0: render this many frames of the sweep without writing files, print one JSON line
    std::size_t bench_warmup = 20;
    std::size_t bench_rounds = 1;
    bool no_solids = false;
    bool print_stats = false;
    bool parse_only = false;      // read the input, generate the solids, report sizes, no GPU work
    bool png = false;             // beside every .vti a colour-mapped .png of one channel (what utility/screen.py makes with ParaView)
    int png_channel = 1;          // 0: tau, 1: I (screen.py:11 colours by component 'Y')
    bool png_fixed_range = false; // false: every frame mapped over its own finite range, as ParaView does on load
    double png_lo = 0, png_hi = 0;
    bool raw_vti = false;         // write uncompressed appended data instead of zlib blocks
    bool reference_algorithm = false;  // force the bin-sort-resolve path (tet soups with overlapping cells)
    bool rccl_selftest = false;   // load librccl, bring up a communicator on --device and run the exchange's call pattern against itself
    std::string selftest_vti;     // write a small synthetic image with the configured encoding and exit (no GPU)
    std::string dump_solids;      // write the generated solid tets (raw doubles) for inspection
    std::size_t frames = 1;       // > 1: sweep, grid stays resident on the GPU
    std::string sweep = "Y";      // which angle advances per frame: X, Y, D or I
    double sweep_step = 1.0 / 180.0;
};

// Process-wide configuration, as in the reference (config.hpp:7-32).
class app {
public:
    static app& instance() {
        static app self;
        return self;
    }
    render_config config{};

private:
    app() = default;
};

constexpr int MAX_NUMBER_OF_THREADS = 32;  // config.hpp:39 (CPU path only; meaningless on the GPU)

constexpr double PI = 3.14159265358979323846;  // config.hpp:45
constexpr double L = 0.945;                    // distance between the stars, R_sol
constexpr double ACC_X0 = 1;                   // accretor position
constexpr double ACC_Y0 = 0;
constexpr double ACC_Z0 = 0;
constexpr double ACC_DISK_R = 0.02;
constexpr double M_ACC = 0.73;
constexpr double M_DONOR = 0.1;
constexpr long double G_SOL = 132700000000000000000.;  // long double on purpose (config.hpp:69)
constexpr double OMEGA = 2 * PI * 10000;

// Image-plane domain {x_max, x_min, y_max, y_min} (main.cpp:83)
constexpr double DOMAIN_BOUNDS[4] = {2.2, -0.2, 0.9, -0.9};
