#include "cli.hpp"

#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "config.hpp"

namespace {

struct option_spec {
    const char* long_name;
    char short_name;  // 0 = none
    bool takes_value;
    const char* help;
    const char* default_text;  // shown as arg (=...) like boost
};

const option_spec kOptions[] = {
    {"help", 'h', false, "produce help message", nullptr},
    {"file", 'f', true, "source file", nullptr},
    {"destination", 'd', true, "destination file", nullptr},
    {"threads", 'j', true, "number of parallel threads", nullptr},
    {"resolution_x", 'x', true, "set x axis resolution", "1200"},
    {"resolution_y", 'y', true, "set y axis resolution", "900"},
    {"angle_around_x", 'X', true, "rotate view plane by angle around x axis", "0"},
    {"angle_around_y", 'Y', true, "rotate view plane by angle around y axis", "0"},
    {"donor_angle", 'D', true, "initial donor angle around y axis", "0"},
    {"initial_system_angle", 'I', true, "initial angle of system y axis", "0"},
    {"alpha_limit", 0, true, "limit alpha value", "2.5"},
    // additions of this build
    {"device", 0, true, "GPU ordinal", "0"},
    {"devices", 0, true, "several GPUs of this node: 0-7 or 0,1,2 (one context per GPU, grid replicated)", nullptr},
    {"exchange", 0, true, "several GPUs, rows: host (each GPU copies its rows to the host image), rccl or p2p", "host"},
    {"split", 0, true, "several GPUs: rows, frames (frame k of a sweep on GPU k mod N) or auto", "auto"},
    {"row_layout", 0, true, "several GPUs, rows: blocks (one contiguous cost-balanced block per GPU), tiles (cyclic) or auto", "auto"},
    {"bench_files", 0, false, "--bench: write every timed frame to <destination>_NNNNN.vti like a sweep (end to end with files)", nullptr},
    {"bench", 0, true, "render this many sweep frames without writing files and print one JSON line", nullptr},
    {"bench_warmup", 0, true, "untimed frames before --bench", "20"},
    {"bench_rounds", 0, true, "--bench: the timed run is repeated this many times; the line carries the whole run, the fastest and the median round", "1"},
    {"auto_bounds", 0, false, "image domain = bounding box of the transformed objects instead of the fixed domain", nullptr},
    {"no_solids", 0, false, "do not generate the Roche lobe and the accretor sphere", nullptr},
    {"stats", 0, false, "print per-stage GPU timings and segment counts", nullptr},
    {"png", 0, false, "also write a colour-mapped PNG (Cool to Warm, NaN yellow) of one channel beside every .vti", nullptr},
    {"png_channel", 0, true, "channel of the PNG: 0 (optical depth) or 1 (intensity)", "1"},
    {"png_range", 0, true, "lo,hi: fixed colour range for every frame (default: each frame's own finite range)", nullptr},
    {"raw_vti", 0, false, "write the .vti uncompressed (default: zlib blocks, like vtkXMLImageDataWriter)", nullptr},
    {"reference_algorithm", 0, false, "bin + sort + resolve on the GPU (for grids with overlapping cells)", nullptr},
    {"rccl_selftest", 0, false, "load librccl, open a communicator on --device and run the tile exchange's calls against itself", nullptr},
    {"selftest_vti", 0, true, "write a synthetic 48x32 image to this .vti and exit (checks the writer, no GPU)", nullptr},
    {"parse_only", 0, false, "read the input and generate the solids, print sizes, no GPU work", nullptr},
    {"dump_solids", 0, true, "write generated solid tets to this file (int64 count + doubles per object)", nullptr},
    {"frames", 0, true, "number of frames of a sweep (grid stays on the GPU)", "1"},
    {"sweep", 0, true, "angle advanced per frame: X, Y, D or I", "Y"},
    {"sweep_step", 0, true, "sweep increment per frame, units of pi", "0.00555556"},
};

const option_spec* find_long(const std::string& name) {
    for (const auto& o : kOptions)
        if (name == o.long_name) return &o;
    return nullptr;
}
const option_spec* find_short(char c) {
    for (const auto& o : kOptions)
        if (o.short_name && o.short_name == c) return &o;
    return nullptr;
}

double to_double(const std::string& opt, const std::string& v) {
    char* end = nullptr;
    const double r = std::strtod(v.c_str(), &end);
    if (v.empty() || *end != '\0')
        throw std::runtime_error("the argument ('" + v + "') for option '--" + opt + "' is invalid");
    return r;
}
long long to_integer(const std::string& opt, const std::string& v) {
    char* end = nullptr;
    const long long r = std::strtoll(v.c_str(), &end, 10);
    if (v.empty() || *end != '\0')
        throw std::runtime_error("the argument ('" + v + "') for option '--" + opt + "' is invalid");
    return r;
}

}  // namespace

void print_usage(std::ostream& out) {
    out << "Allowed options:\n";
    for (const auto& o : kOptions) {
        std::string left = "  ";
        if (o.short_name) {
            left += "-";
            left += o.short_name;
            left += " [ --";
            left += o.long_name;
            left += " ]";
        } else {
            left += "--";
            left += o.long_name;
        }
        if (o.takes_value) {
            left += " arg";
            if (o.default_text) left += std::string(" (=") + o.default_text + ")";
        }
        if (left.size() < 41) left.resize(41, ' ');
        else left += " ";
        out << left << o.help << "\n";
    }
    out << std::endl;
}

bool program_options(int argc, char** argv, std::ostream& out) {
    render_config& cfg = app::instance().config;
    bool want_help = false, have_file = false, have_dest = false, have_threads = false;

    auto apply = [&](const option_spec& o, const std::string& v) {
        const std::string n = o.long_name;
        if (n == "help") want_help = true;
        else if (n == "file") { cfg.file = v; have_file = true; }
        else if (n == "destination") { cfg.destination = v; have_dest = true; }
        else if (n == "threads") { cfg.threads = static_cast<int>(to_integer(n, v)); have_threads = true; }
        else if (n == "resolution_x") cfg.resolution_x = static_cast<std::size_t>(to_integer(n, v));
        else if (n == "resolution_y") cfg.resolution_y = static_cast<std::size_t>(to_integer(n, v));
        else if (n == "angle_around_x") cfg.angle_around_x = to_double(n, v);
        else if (n == "angle_around_y") cfg.angle_around_y = to_double(n, v);
        else if (n == "donor_angle") cfg.donor_angle = to_double(n, v);
        else if (n == "initial_system_angle") cfg.system_initial_angle_around_y = to_double(n, v);
        else if (n == "alpha_limit") cfg.limit_alpha_value = to_double(n, v);
        else if (n == "device") cfg.device = static_cast<int>(to_integer(n, v));
        else if (n == "devices") cfg.devices = v;
        else if (n == "exchange") cfg.exchange = v;
        else if (n == "split") cfg.split = v;
        else if (n == "row_layout") cfg.row_layout = v;
        else if (n == "bench_files") cfg.bench_files = true;
        else if (n == "bench") cfg.bench = static_cast<std::size_t>(std::max(0ll, to_integer(n, v)));
        else if (n == "bench_warmup") cfg.bench_warmup = static_cast<std::size_t>(std::max(0ll, to_integer(n, v)));
        else if (n == "bench_rounds") cfg.bench_rounds = static_cast<std::size_t>(std::max(1ll, to_integer(n, v)));
        else if (n == "auto_bounds") cfg.auto_bounds = true;
        else if (n == "no_solids") cfg.no_solids = true;
        else if (n == "stats") cfg.print_stats = true;
        else if (n == "raw_vti") cfg.raw_vti = true;
        else if (n == "png") cfg.png = true;
        else if (n == "png_channel") {
            cfg.png_channel = static_cast<int>(to_integer(n, v));
            if (cfg.png_channel != 0 && cfg.png_channel != 1)
                throw std::runtime_error("the argument ('" + v + "') for option '--png_channel' is invalid");
        } else if (n == "png_range") {
            const auto comma = v.find(',');
            if (comma == std::string::npos) throw std::runtime_error("the argument ('" + v + "') for option '--png_range' is invalid");
            cfg.png_lo = to_double(n, v.substr(0, comma));
            cfg.png_hi = to_double(n, v.substr(comma + 1));
            cfg.png_fixed_range = true;
        }
        else if (n == "reference_algorithm") cfg.reference_algorithm = true;
        else if (n == "selftest_vti") cfg.selftest_vti = v;
        else if (n == "rccl_selftest") cfg.rccl_selftest = true;
        else if (n == "parse_only") cfg.parse_only = true;
        else if (n == "dump_solids") cfg.dump_solids = v;
        else if (n == "frames") cfg.frames = static_cast<std::size_t>(std::max(1ll, to_integer(n, v)));
        else if (n == "sweep") cfg.sweep = v;
        else if (n == "sweep_step") cfg.sweep_step = to_double(n, v);
    };

    for (int i = 1; i < argc; ++i) {
        const std::string tok = argv[i];
        const option_spec* spec = nullptr;
        std::string value;
        bool have_value = false;
        if (tok.rfind("--", 0) == 0 && tok.size() > 2) {
            const auto eq = tok.find('=');
            const std::string name = tok.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            spec = find_long(name);
            if (!spec) throw std::runtime_error("unrecognised option '" + tok + "'");
            if (eq != std::string::npos) {
                value = tok.substr(eq + 1);
                have_value = true;
            }
        } else if (tok.size() >= 2 && tok[0] == '-' && tok != "--") {
            spec = find_short(tok[1]);
            if (!spec) throw std::runtime_error("unrecognised option '" + tok + "'");
            if (tok.size() > 2) {  // attached value: -j16 (readme.md:40)
                value = tok.substr(2);
                have_value = true;
            }
        } else {
            throw std::runtime_error("too many positional options have been specified on the command line");
        }
        if (spec->takes_value && !have_value) {
            if (i + 1 >= argc)
                throw std::runtime_error(std::string("the required argument for option '--") + spec->long_name + "' is missing");
            value = argv[++i];
        } else if (!spec->takes_value && have_value) {
            throw std::runtime_error(std::string("option '--") + spec->long_name + "' does not take any arguments");
        }
        apply(*spec, value);
    }

    if (want_help) {  // main.cpp:38-41
        print_usage(out);
        return false;
    }
    if (!cfg.selftest_vti.empty() || cfg.rccl_selftest) return true;
    if (cfg.bench > 0 && !cfg.bench_files && have_file && !have_dest) have_dest = true;  // a benchmark writes no file
    if (cfg.exchange != "host" && cfg.exchange != "rccl" && cfg.exchange != "p2p")
        throw std::runtime_error("the argument ('" + cfg.exchange + "') for option '--exchange' is invalid");
    if (cfg.split != "auto" && cfg.split != "rows" && cfg.split != "frames")
        throw std::runtime_error("the argument ('" + cfg.split + "') for option '--split' is invalid");
    if (cfg.row_layout != "auto" && cfg.row_layout != "blocks" && cfg.row_layout != "tiles")
        throw std::runtime_error("the argument ('" + cfg.row_layout + "') for option '--row_layout' is invalid");
    if (!(have_file && have_dest)) {  // main.cpp:43-51
        out << "Error! Source filename and destination filename must be specified" << std::endl;
        print_usage(out);
        return false;
    }
    if (!have_threads)  // main.cpp:54-58
        cfg.threads = std::max(static_cast<int>(std::thread::hardware_concurrency()), 1);
    return true;
}
