// Sanitizer harness for the CPU side: the host-only translation units of the product
// (csrc/host/vtk_io.cpp, cli.cpp, scene.cpp; csrc/adjacency.cpp) and the oracle (oracle/oracle.cpp) built with
// -fsanitize=address,undefined into ONE program with no HIP runtime and no GPU.  tests/test_sanitizers_cpu.py builds it
// and drives it with well-formed and malformed inputs: every run must end in a result or an error MESSAGE, never in
// a sanitizer report.
//
//   host_san read   <file.vtk>              parse (tokenizer, big-endian, 5.1 layout), weld, adjacency -> counts
//   host_san vti    <out.vti> <w> <h> <raw> write a synthetic frame (zero blocks, NaNs, ragged last block) + PNG
//   host_san cli    <args...>               the option parser on the given command line
//   host_san solids                         init_polar: Roche lobe + sphere, unique faces -> counts
//   host_san oracle <scene.bin> <out.f32>   the CPU oracle on a dumped scene (see the pytest for the layout)
//   host_san blocks <world> <base> <c0> <c1> ...  cost-balanced row blocks of the native multi-GPU host
//   host_san wblocks <world> <base> <t0> .. <c0> <c1> ...  the same blocks cut again by the times t_k their devices took
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <limits>
#include <stdexcept>
#include <string>
#include <random>
#include <zlib.h>
#include <vector>

#include "adjacency.hpp"
#include "cli.hpp"
#include "config.hpp"
#include "row_blocks.hpp"
#include <cstdlib>
#include "scene.hpp"
#include "fast_deflate.hpp"
#include "vtk_io.hpp"

extern "C" int c5o_render(const double* xyz, int64_t n_pts, const int32_t* cell_vert, int64_t n_cells, const double* alpha,
                          const double* q, const double* rots, int n_rot, const double* solid_tets, const double* solid_colour,
                          int64_t n_solid, int res_x, int res_y, const double* bounds4, double alpha_limit, int threads, float* out,
                          int64_t* stats, double* timing_ms, const int32_t* probe_ij, int n_probe, int probe_cap, double* probe_out,
                          int32_t* probe_count, char* err, int errlen);

namespace {
template <class T>
std::vector<T> take(std::ifstream& f, size_t n) {
    std::vector<T> v(n);
    f.read(reinterpret_cast<char*>(v.data()), static_cast<std::streamsize>(n * sizeof(T)));
    if (!f) throw std::runtime_error("scene file too short");
    return v;
}
}  // namespace

int main(int argc, char** argv) try {
    if (argc < 2) throw std::runtime_error("usage: host_san read|vti|cli|solids|oracle ...");
    const std::string mode = argv[1];
    if (mode == "read") {
        if (argc < 3) throw std::runtime_error("read: file?");
        const vtk_grid g = read_legacy_vtk(argv[2]);
        std::vector<int32_t> rep, adj;
        std::vector<uint32_t> bfaces;
        std::string err;
        const int64_t merged = c5::weld_points(g.points.data(), g.n_points(), rep);
        std::vector<int32_t> cells(g.tets);
        for (int32_t& id : cells) {
            if (id < 0 || id >= g.n_points()) throw std::runtime_error("cell references a point id out of range");
            id = rep[static_cast<size_t>(id)];
        }
        const bool ok = c5::build_face_adjacency(cells.data(), g.n_cells(), g.n_points(), adj, bfaces, err);
        std::printf("points %lld cells %lld scalars %zu merged %lld conforming %d boundary_faces %zu%s%s\n",
                    static_cast<long long>(g.n_points()), static_cast<long long>(g.n_cells()), g.cell_scalars.size(),
                    static_cast<long long>(merged), ok ? 1 : 0, bfaces.size(), ok ? "" : " : ", ok ? "" : err.c_str());
        // the same through the scene object the CLI builds (object3d_base::read_vtk_file)
        object3d_accretion_disk disk{std::string(argv[2])};
        std::printf("disk cells %zu\n", disk.get_pointer()->size());
        return 0;
    }
    if (mode == "vti") {
        if (argc < 6) throw std::runtime_error("vti: out w h raw?");
        const int w = std::atoi(argv[3]), h = std::atoi(argv[4]);
        const bool raw = std::atoi(argv[5]) != 0;
        std::vector<float> px(static_cast<size_t>(w) * h * 2, 0.0f);
        for (int y = h / 3; y < 2 * h / 3; ++y)  // a band of values between bands of zeros (whole zero blocks either side)
            for (int x = w / 4; x < 3 * w / 4; ++x) {
                px[(static_cast<size_t>(y) * w + x) * 2] = static_cast<float>(x) * 0.25f + static_cast<float>(y);
                px[(static_cast<size_t>(y) * w + x) * 2 + 1] = std::sin(static_cast<float>(x * y));
            }
        px[0] = -0.0f;  // not a zero block: -0.0 has a bit set
        px[px.size() - 1] = std::numeric_limits<float>::quiet_NaN();
        write_vti(argv[2], px.data(), w, h, !raw);
        double lo = 0, hi = 0;
        colour_range(px.data(), w, h, 1, &lo, &hi);
        write_png(std::string(argv[2]) + ".png", px.data(), w, h, 1, lo, hi);
        std::printf("wrote %dx%d range %g %g\n", w, h, lo, hi);
        return 0;
    }
    if (mode == "deflate") {
        // fast_deflate.cpp against zlib's own inflate: blocks of widened floats of every kind (smooth, noisy, runs of one
        // value, specials, short blocks), and doubles that are NOT widened floats (must be declined)
        const int rounds = argc > 2 ? std::atoi(argv[2]) : 200;
        std::mt19937 rng(12345);
        std::vector<double> vals;
        std::vector<unsigned char> packed, back;
        size_t total_in = 0, total_out = 0, total_zlib = 0;
        for (int r = 0; r < rounds; ++r) {
            const size_t n = (r % 7 == 0) ? 2 + rng() % 40 : (r % 5 == 0 ? 4096 : 1 + rng() % 4096);
            vals.resize(n);
            const int kind = r % 6;
            float walk = 0.0f;
            for (size_t i = 0; i < n; ++i) {
                float v;
                switch (kind) {
                case 0: v = std::ldexp(static_cast<float>(rng() & 0xFFFFFF), -20 - static_cast<int>(rng() % 8)); break;       // noisy mantissas
                case 1: walk += 0.01f * static_cast<float>(static_cast<int>(rng() % 200) - 100); v = walk; break;           // smooth
                case 2: v = (rng() % 9 == 0) ? static_cast<float>(rng() % 5) : 0.0f; break;                                  // mostly zeros
                case 3: v = (i / 37 % 2) ? std::numeric_limits<float>::quiet_NaN() : 7.5f; break;                            // runs
                case 4: { uint32_t b = rng(); std::memcpy(&v, &b, 4); break; }                                               // any bit pattern (NaNs, denormals, infinities)
                default: v = (i % 3 == 0) ? -0.0f : (i % 3 == 1 ? std::numeric_limits<float>::infinity() : 1e-42f); break;
                }
                vals[i] = static_cast<double>(v);
            }
            const size_t cap = compressBound(static_cast<uLong>(8 * n));
            packed.assign(cap, 0xAB);
            const size_t sz = c5::deflate_widened_doubles(vals.data(), n, packed.data(), cap);
            if (n < 2) {
                if (sz != 0) throw std::runtime_error("a single value must be declined");
                continue;
            }
            if (sz == 0) {
                // (a short block may not hold the code tables within compressBound, random bits may not compress: allowed)
                if (kind != 4 && kind != 0 && n >= 64) throw std::runtime_error("declined a block it should take, kind " + std::to_string(kind));
                continue;
            }
            {   // the float entry point writes the same stream
                std::vector<float> fl(n);
                for (size_t i = 0; i < n; ++i) fl[i] = static_cast<float>(vals[i]);
                std::vector<unsigned char> again(cap, 0xEF);
                bool same_input = true;
                for (size_t i = 0; i < n; ++i) {  // (a NaN's payload survives the round trip float -> double -> float on this target)
                    const double d = static_cast<double>(fl[i]);
                    same_input = same_input && std::memcmp(&d, &vals[i], 8) == 0;
                }
                const size_t sz2 = c5::deflate_floats_as_doubles(fl.data(), n, again.data(), cap);
                if (same_input && (sz2 != sz || std::memcmp(again.data(), packed.data(), sz) != 0)) throw std::runtime_error("the float entry point differs");
            }
            back.assign(8 * n, 0);
            uLongf got = static_cast<uLongf>(back.size());
            const int rc = uncompress(back.data(), &got, packed.data(), static_cast<uLong>(sz));
            if (rc != Z_OK || got != 8 * n || std::memcmp(back.data(), vals.data(), 8 * n) != 0)
                throw std::runtime_error("round trip failed: rc " + std::to_string(rc) + " kind " + std::to_string(kind) + " n " + std::to_string(n));
            uLongf zs = static_cast<uLongf>(cap);
            std::vector<unsigned char> z(cap);
            compress2(z.data(), &zs, reinterpret_cast<const Bytef*>(vals.data()), static_cast<uLong>(8 * n), Z_BEST_SPEED);
            total_in += 8 * n;
            total_out += sz;
            total_zlib += zs;
        }
        // not widened floats: declined
        std::vector<double> other(100);
        for (size_t i = 0; i < other.size(); ++i) other[i] = 1.0 / (3.0 + static_cast<double>(i));
        std::vector<unsigned char> o(4096);
        if (c5::deflate_widened_doubles(other.data(), other.size(), o.data(), o.size()) != 0) throw std::runtime_error("took doubles that are not widened floats");
        // a buffer that is too small: declined, nothing written past it
        std::vector<double> big(4096, 0.0);
        for (size_t i = 0; i < big.size(); ++i) big[i] = static_cast<double>(static_cast<float>(rng()) * 1e-3f);
        std::vector<unsigned char> small(1000 + 16, 0xCD);
        if (c5::deflate_widened_doubles(big.data(), big.size(), small.data(), 1000) != 0) throw std::runtime_error("claimed to fit 1000 bytes");
        for (size_t i = 1000; i < small.size(); ++i)
            if (small[i] != 0xCD) throw std::runtime_error("wrote past the end of its buffer");
        std::printf("deflate ok: %zu bytes in, %zu out (zlib level 1: %zu)\n", total_in, total_out, total_zlib);
        return 0;
    }
    if (mode == "deflate_file") {  // raw float32 file: sizes and single-thread times of both compressors, 32 KB blocks of doubles
        if (argc < 3) throw std::runtime_error("deflate_file: path");
        std::ifstream f(argv[2], std::ios::binary);
        std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        const size_t n = raw.size() / 4;
        std::vector<double> vals(n);
        for (size_t i = 0; i < n; ++i) {
            float v;
            std::memcpy(&v, raw.data() + 4 * i, 4);
            vals[i] = static_cast<double>(v);
        }
        const size_t cap = compressBound(32768);
        std::vector<unsigned char> out(cap);
        size_t fast = 0, slow = 0, declined = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < n; i += 4096) {
            const size_t sz = c5::deflate_widened_doubles(vals.data() + i, std::min<size_t>(4096, n - i), out.data(), cap);
            fast += sz;
            declined += sz == 0;
        }
        auto t1 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < n; i += 4096) {
            uLongf zs = static_cast<uLongf>(cap);
            compress2(out.data(), &zs, reinterpret_cast<const Bytef*>(vals.data() + i), static_cast<uLong>(8 * std::min<size_t>(4096, n - i)), Z_BEST_SPEED);
            slow += zs;
        }
        auto t2 = std::chrono::steady_clock::now();
        std::printf("%zu doubles: fast %zu bytes in %.1f ms (%zu blocks declined), zlib level 1 %zu bytes in %.1f ms\n", n, fast,
                    std::chrono::duration<double, std::milli>(t1 - t0).count(), declined, slow, std::chrono::duration<double, std::milli>(t2 - t1).count());
        return 0;
    }
    if (mode == "vti_time") {  // raw float32 image file, w, h, repeats: time of write_vti (zlib .vti) per call
        if (argc < 6) throw std::runtime_error("vti_time: in.f32 w h repeats");
        const int w = std::atoi(argv[3]), h = std::atoi(argv[4]), reps = std::atoi(argv[5]);
        std::ifstream f(argv[2], std::ios::binary);
        std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        if (raw.size() != static_cast<size_t>(w) * h * 8) throw std::runtime_error("vti_time: size");
        const std::string out = std::string(argv[2]) + ".vti";
        write_vti(out, reinterpret_cast<const float*>(raw.data()), w, h, true);
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) write_vti(out, reinterpret_cast<const float*>(raw.data()), w, h, true);
        const auto t1 = std::chrono::steady_clock::now();
        std::ifstream g(out, std::ios::binary | std::ios::ate);
        std::printf("write_vti %dx%d: %.2f ms per call, file %lld bytes\n", w, h, std::chrono::duration<double, std::milli>(t1 - t0).count() / reps,
                    static_cast<long long>(g.tellg()));
        return 0;
    }
    if (mode == "cli") {
        const bool go = program_options(argc - 1, argv + 1, std::cout);
        const render_config& c = app::instance().config;
        std::printf("go %d file '%s' res %zux%zu frames %zu devices '%s' layout %s\n", go ? 1 : 0, c.file.c_str(), c.resolution_x,
                    c.resolution_y, c.frames, c.devices.c_str(), c.row_layout.c_str());
        return 0;
    }
    if (mode == "solids") {
        object3d_roche_lobe lobe(point{ACC_X0, ACC_Y0, ACC_Z0}, L, 0.3 * PI, M_ACC, M_DONOR, OMEGA);
        object3d_sphere sphere(point{ACC_X0, ACC_Y0, ACC_Z0}, ACC_DISK_R);
        for (object3d_base* o : {static_cast<object3d_base*>(&lobe), static_cast<object3d_base*>(&sphere)}) {
            const object3d_data& d = *o->get_pointer();
            std::vector<double> pts;
            std::vector<int32_t> faces;
            c5::unique_solid_faces(d.soup.data(), static_cast<int64_t>(d.soup.size() / 12), pts, faces);
            std::printf("solid cells %zu unique points %zu unique faces %zu\n", d.size(), pts.size() / 3, faces.size() / 4);
        }
        return 0;
    }
    if (mode == "blocks") {
        if (argc < 5) throw std::runtime_error("blocks: world base costs...");
        std::vector<uint32_t> cost;
        for (int k = 4; k < argc; ++k) cost.push_back(static_cast<uint32_t>(std::strtoul(argv[k], nullptr, 10)));
        const char* q = std::getenv("C5_BLOCK_QUANTUM");
        for (const auto& b : balanced_row_blocks(cost, std::atoi(argv[2]), std::atof(argv[3]), q ? std::atoi(q) : 1)) std::printf("%d %d\n", b.first, b.second);
        return 0;
    }
    if (mode == "wblocks") {  // blocks cut again by measured times: world base t0 .. t(world-1) costs...
        if (argc < 5) throw std::runtime_error("wblocks: world base times... costs...");
        const int world = std::atoi(argv[2]);
        const double base = std::atof(argv[3]);
        if (argc < 4 + world + world) throw std::runtime_error("wblocks: too few numbers");
        std::vector<double> times;
        for (int k = 0; k < world; ++k) times.push_back(std::atof(argv[4 + k]));
        std::vector<uint32_t> cost;
        for (int k = 4 + world; k < argc; ++k) cost.push_back(static_cast<uint32_t>(std::strtoul(argv[k], nullptr, 10)));
        const auto first = balanced_row_blocks(cost, world, base);
        for (const auto& b : balanced_row_blocks(time_weighted_row_costs(cost, base, first, times), world)) std::printf("%d %d\n", b.first, b.second);
        return 0;
    }
    if (mode == "oracle") {
        if (argc < 4) throw std::runtime_error("oracle: scene out?");
        std::ifstream f(argv[2], std::ios::binary);
        if (!f) throw std::runtime_error("cannot open scene");
        const std::vector<int64_t> hd = take<int64_t>(f, 6);  // n_pts, n_cells, n_rot, res_x, res_y, n_solid
        const std::vector<double> xyz = take<double>(f, static_cast<size_t>(3 * hd[0]));
        const std::vector<int32_t> cells = take<int32_t>(f, static_cast<size_t>(4 * hd[1]));
        const std::vector<double> alpha = take<double>(f, static_cast<size_t>(hd[1])), q = take<double>(f, static_cast<size_t>(hd[1]));
        const std::vector<double> rots = take<double>(f, static_cast<size_t>(3 * hd[2]));
        const std::vector<double> bounds = take<double>(f, 4), limit = take<double>(f, 1);
        const std::vector<double> solids = take<double>(f, static_cast<size_t>(12 * hd[5])), colour = take<double>(f, static_cast<size_t>(hd[5]));
        std::vector<float> out(static_cast<size_t>(hd[3] * hd[4] * 2));
        int64_t stats[4] = {0, 0, 0, 0};
        double timing[3];
        char err[512] = {0};
        const int rc = c5o_render(xyz.data(), hd[0], cells.data(), hd[1], alpha.data(), q.data(), rots.data(), static_cast<int>(hd[2]),
                                  hd[5] ? solids.data() : nullptr, hd[5] ? colour.data() : nullptr, hd[5], static_cast<int>(hd[3]),
                                  static_cast<int>(hd[4]), bounds.data(), limit[0], 2, out.data(), stats, timing, nullptr, 0, 0, nullptr,
                                  nullptr, err, sizeof err);
        if (rc != 0) throw std::runtime_error(std::string("oracle: ") + err);
        std::ofstream o(argv[3], std::ios::binary);
        o.write(reinterpret_cast<const char*>(out.data()), static_cast<std::streamsize>(out.size() * sizeof(float)));
        std::printf("segments %lld covered %lld marked %lld\n", static_cast<long long>(stats[0]), static_cast<long long>(stats[1]),
                    static_cast<long long>(stats[2]));
        return 0;
    }
    throw std::runtime_error("unknown mode '" + mode + "'");
} catch (const std::exception& e) {
    std::fprintf(stderr, "host_san: %s\n", e.what());
    return 1;
}
