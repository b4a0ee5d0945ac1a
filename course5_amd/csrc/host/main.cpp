// `course` — command line of the MI355X build.  Same options, banner, timing lines, input and
// output contract as the reference's project/src/main.cpp; the per-pixel OpenMP loops behind
// `plane` are replaced by the HIP kernels in libcourse5_hip.so.
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <deque>
#include <memory>
#include <iostream>
#include <limits>
#include <stdexcept>
#include <thread>

#include "cli.hpp"
#include "config.hpp"
#include "plane.hpp"
#include "scene.hpp"

namespace {
const auto& timestamp = std::chrono::high_resolution_clock::now;

long long ms_between(std::chrono::high_resolution_clock::time_point a, std::chrono::high_resolution_clock::time_point b) {
    return std::chrono::duration_cast<std::chrono::milliseconds>(b - a).count();
}

std::string frame_name(const std::string& destination, std::size_t k, std::size_t frames) {
    if (frames <= 1) return destination;
    const auto dot = destination.rfind('.');
    char tag[32];
    std::snprintf(tag, sizeof tag, "_%05zu", k);
    return dot == std::string::npos ? destination + tag : destination.substr(0, dot) + tag + destination.substr(dot);
}

std::string png_name(const std::string& vti) {
    const auto dot = vti.rfind('.');
    const auto slash = vti.rfind('/');
    const bool has_ext = dot != std::string::npos && (slash == std::string::npos || dot > slash);
    return (has_ext ? vti.substr(0, dot) : vti) + ".png";
}

void write_frame(const object2d& img, const std::string& name) {
    img.export_to_vti(name);
    if (app::instance().config.png) img.export_to_png(png_name(name));
}
}  // namespace

int main(int argc, char** argv) try {
    const render_config& config = app::instance().config;
    if (!program_options(argc, argv, std::cout)) return 0;  // main.cpp:74-78

    if (config.rccl_selftest) {
        std::cout << rccl_selftest(config.device) << std::endl;
        return 0;
    }
    if (!config.selftest_vti.empty()) {  // writer check without a GPU: value = x + 100 y (+0.5 on ch1), one NaN pixel
        const int w = 48, h = 32;
        std::vector<float> px(static_cast<size_t>(w) * h * 2);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                px[(static_cast<size_t>(y) * w + x) * 2] = static_cast<float>(x + 100 * y);
                px[(static_cast<size_t>(y) * w + x) * 2 + 1] = static_cast<float>(x + 100 * y) + 0.5f;
            }
        px[(5 * w + 7) * 2] = px[(5 * w + 7) * 2 + 1] = std::numeric_limits<float>::quiet_NaN();
        write_frame(object2d(px, w, h), config.selftest_vti);
        return 0;
    }

    // main.cpp:83; --auto_bounds: empty, so that plane computes the objects' bounding box (plane.cpp:278-288)
    const std::vector<double> domain = config.auto_bounds ? std::vector<double>{} : std::vector<double>(DOMAIN_BOUNDS, DOMAIN_BOUNDS + 4);

    // main.cpp:85-92
    std::cout << "Defined grid resolution: " << config.resolution_x << "x" << config.resolution_y << std::endl;
    std::cout << "Source file: " << config.file << std::endl;
    std::cout << "Number of parallel threads: " << config.threads << std::endl;
    std::cout << "Initial rotate angle of roche lobe: " << config.donor_angle << " Pi" << std::endl;
    std::cout << "Plane angle around x: " << config.angle_around_x << " Pi" << std::endl;
    std::cout << "Plane angle around y: " << config.angle_around_y << " Pi" << std::endl;
    std::cout << "Initial system angle around y: " << config.system_initial_angle_around_y << " Pi" << std::endl;
    std::cout << "Limit alpha value: " << config.limit_alpha_value << std::endl;

    // main.cpp:125: -j sizes the OpenMP team of the host-side work (here: adjacency build, .vti writer)
    omp_set_num_threads(static_cast<int>(std::max<std::size_t>(1, config.threads)));

    render_config view = config;  // angles that a sweep advances
    auto apply_view = [&](object3d_base& disk, object3d_base* lobe) {
        // main.cpp:96,105-107 and 112-114; the lobe first turns by the donor angle (object3d_roche_lobe.cpp:48)
        const double make_perpendicular_to_y_angle = -view.system_initial_angle_around_y * PI + PI / 2.;
        auto three = [&](object3d_base& o) {
            o.rotate_around_x_axis(make_perpendicular_to_y_angle);
            o.rotate_around_y_axis(view.angle_around_y * PI, ACC_X0);
            o.rotate_around_x_axis(-make_perpendicular_to_y_angle + view.angle_around_x * PI);
        };
        disk.clear_rotations();
        three(disk);
        if (lobe) {
            lobe->clear_rotations();
            lobe->rotate_around_y_axis(view.donor_angle * PI, ACC_X0);
            three(*lobe);
        }
    };

    auto t1 = timestamp();
    object3d_accretion_disk acc_disk{};
    std::string load_error;
    std::thread acc_t([&]() {  // the file is parsed while the solids are generated (main.cpp:98-108)
        try {
            acc_disk = object3d_accretion_disk{config.file};
        } catch (const std::exception& e) {
            load_error = e.what();
        }
    });
    std::vector<object3d_base> objects;
    std::unique_ptr<object3d_roche_lobe> roche_lobe;
    std::unique_ptr<object3d_sphere> acc_sphere;
    if (!config.no_solids) {
        roche_lobe = std::make_unique<object3d_roche_lobe>(point{ACC_X0, ACC_Y0, ACC_Z0}, L, config.donor_angle * PI,
                                                           M_ACC, M_DONOR, OMEGA);
        acc_sphere = std::make_unique<object3d_sphere>(point{ACC_X0, ACC_Y0, ACC_Z0}, ACC_DISK_R);  // never rotated (main.cpp:116)
    }
    acc_t.join();
    if (!load_error.empty()) throw std::runtime_error(load_error);
    apply_view(acc_disk, roche_lobe.get());
    auto t2 = timestamp();
    std::cout << "Loading data with VTK lib and other preparations completed in " << ms_between(t1, t2) << " ms. "
              << std::endl;

    if (!config.dump_solids.empty()) {
        std::FILE* f = std::fopen(config.dump_solids.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot write '" + config.dump_solids + "'");
        for (object3d_base* o : {static_cast<object3d_base*>(roche_lobe.get()), static_cast<object3d_base*>(acc_sphere.get())}) {
            if (!o) continue;
            const object3d_data& d = *o->get_pointer();
            const long long n = static_cast<long long>(d.soup.size() / 12);
            std::fwrite(&n, sizeof n, 1, f);
            std::fwrite(d.soup.data(), sizeof(double), d.soup.size(), f);
        }
        std::fclose(f);
    }
    if (config.parse_only) {
        const object3d_data& d = *acc_disk.get_pointer();
        double sa = 0, sq = 0;
        for (double v : d.value0) sa += v;
        for (double v : d.value1) sq += v;
        std::printf("grid: %zu points, %zu cells, sum(alpha) = %.17g, sum(Q) = %.17g\n", d.points.size() / 3, d.size(), sa, sq);
        if (roche_lobe) std::printf("roche lobe: %zu solid cells\n", roche_lobe->get_pointer()->size());
        if (acc_sphere) std::printf("sphere: %zu solid cells\n", acc_sphere->get_pointer()->size());
        return 0;
    }

    objects.push_back(acc_disk);
    if (roche_lobe) objects.push_back(*roche_lobe);
    if (acc_sphere) objects.push_back(*acc_sphere);

    // GPUs: one (the reference's flow as it is) or several of this node.  Several GPUs either split every
    // frame by rows (config 4: one image, cyclic 16-row tiles) or take whole frames of a sweep in turn
    // (config 5: frame k on GPU k mod N, nothing to exchange) — frames are independent, rows of a frame too
    // (plane.cpp:161-169), and the grid is small enough to live on every GPU.
    const std::vector<int> devices = config.devices.empty() ? std::vector<int>{config.device} : parse_device_list(config.devices);
    const exchange_mode exchange = config.exchange == "rccl" ? exchange_mode::rccl
                                   : config.exchange == "p2p" ? exchange_mode::p2p : exchange_mode::host;
    const bool is_sweep = config.frames > 1 || config.bench > 0;
    const bool by_frames = devices.size() > 1 && (config.split == "frames" || (config.split == "auto" && is_sweep));

    t1 = timestamp();
    std::vector<std::unique_ptr<plane>> planes;
    if (by_frames) {
        planes.resize(devices.size());
        std::vector<std::thread> makers;
        std::vector<std::string> errors(devices.size());
        for (std::size_t d = 0; d < devices.size(); ++d)
            makers.emplace_back([&, d]() {  // grid upload + adjacency per GPU, side by side
                try {
                    planes[d] = std::make_unique<plane>(config.resolution_x, config.resolution_y, objects, domain,
                                                        std::vector<int>{devices[d]}, exchange_mode::host);
                } catch (const std::exception& e) {
                    errors[d] = e.what();
                }
            });
        for (std::thread& t : makers) t.join();
        for (const std::string& e : errors)
            if (!e.empty()) throw std::runtime_error(e);
    } else {
        planes.push_back(std::make_unique<plane>(config.resolution_x, config.resolution_y, objects, domain, devices, exchange));
    }
    plane& base_plane = *planes[0];
    base_plane.find_intersections();
    object2d result = base_plane.trace_rays(tetra_value::alpha, tetra_value::Q);
    t2 = timestamp();
    std::cout << "Ray-tracing completed in " << ms_between(t1, t2) << " ms. " << std::endl;  // main.cpp:131-135

    if (config.bench == 0) write_frame(result, frame_name(config.destination, 0, config.frames));
    if (config.print_stats) {
        const c5_stats st = base_plane.stats();
        std::cout << "GPU frame: " << st.ms_total << " ms (transform " << st.ms_transform << ", records " << st.ms_records
                  << ", entries " << st.ms_entries << ", solids " << st.ms_solids << ", walk " << st.ms_walk << "); "
                  << st.segments << " segments, " << st.covered_pixels << " covered pixels, " << st.solid_pixels
                  << " solid pixels" << std::endl;
    }

    // sweep: the grid, its adjacency and the solids stay on the GPU(s); only rotation lists change.  Frames are
    // issued ahead (find_intersections returns at once) and retired in order: while frame k is being written,
    // frames k + 1 ... are rendered and copied to pinned host images.
    struct pending_frame {
        std::size_t k;
        plane* p;
    };
    std::deque<pending_frame> in_flight;
    const std::size_t per_plane = (by_frames || exchange == exchange_mode::host) ? 2 : 1;
    const std::size_t max_in_flight = per_plane * planes.size();
    double* const swept = config.sweep == "X"   ? &view.angle_around_x
                          : config.sweep == "D" ? &view.donor_angle
                          : config.sweep == "I" ? &view.system_initial_angle_around_y
                                                : &view.angle_around_y;
    auto issue = [&](std::size_t k, bool advance) {
        if (advance) *swept += config.sweep_step;
        apply_view(objects[0], roche_lobe ? &objects[1] : nullptr);
        plane* p = planes[k % planes.size()].get();
        p->update_views(objects);
        p->find_intersections();
        in_flight.push_back({k, p});
    };
    auto retire = [&](bool write) {
        const pending_frame f = in_flight.front();
        in_flight.pop_front();
        object2d img = f.p->trace_rays(tetra_value::alpha, tetra_value::Q);
        if (write) write_frame(img, frame_name(config.destination, f.k, config.frames));
    };
    if (config.bench == 0) {
        for (std::size_t k = 1; k < config.frames; ++k) {
            while (in_flight.size() >= max_in_flight) retire(true);
            issue(k, true);
        }
        while (!in_flight.empty()) retire(true);
        if (config.frames > 1) {
            const auto t3 = timestamp();
            std::cout << config.frames - 1 << " further frames in " << ms_between(t2, t3) << " ms. " << std::endl;
        }
        if (config.print_stats) {
            std::size_t again = 0;
            for (const auto& p : planes) again += p->retries();
            std::cout << "Frames rendered again after an internal buffer grew: " << again << std::endl;
        }
    } else {
        // --bench: frames rendered and delivered to host memory, no files.  One JSON line.
        for (std::size_t k = 0; k < config.bench_warmup; ++k) {
            while (in_flight.size() >= max_in_flight) retire(false);
            issue(k, false);
        }
        while (!in_flight.empty()) retire(false);
        std::size_t retries0 = 0;
        for (const auto& p : planes) retries0 += p->retries();
        const auto b0 = std::chrono::steady_clock::now();
        for (std::size_t k = 0; k < config.bench; ++k) {
            while (in_flight.size() >= max_in_flight) retire(false);
            issue(k, config.sweep_step != 0.0);
        }
        while (!in_flight.empty()) retire(false);
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - b0).count();
        std::size_t retries1 = 0;
        for (const auto& p : planes) retries1 += p->retries();
        const double rays = static_cast<double>(config.resolution_x) * static_cast<double>(config.resolution_y);
        std::printf("{\"course_bench\": {\"frames\": %zu, \"warmup\": %zu, \"ms_per_frame\": %.4f, \"mrays_per_s\": %.1f, "
                    "\"res_x\": %zu, \"res_y\": %zu, \"n_devices\": %zu, \"split\": \"%s\", \"exchange\": \"%s\", "
                    "\"sweep\": \"%s\", \"sweep_step\": %g, \"solids\": %s, \"retries\": %zu, "
                    "\"delivered_to\": \"pinned host memory\"}}\n",
                    config.bench, config.bench_warmup, secs * 1e3 / static_cast<double>(config.bench),
                    rays * static_cast<double>(config.bench) / secs / 1e6, config.resolution_x, config.resolution_y, devices.size(),
                    devices.size() == 1 ? "none" : (by_frames ? "frames" : "rows"), by_frames ? "none" : config.exchange.c_str(),
                    config.sweep.c_str(), config.sweep_step, config.no_solids ? "false" : "true", retries1 - retries0);
        std::fflush(stdout);
    }
    std::cout << "Result exported. Calculations completed." << std::endl;
    return 0;
} catch (const std::exception& e) {
    // the reference lets std::runtime_error escape and abort; report and fail instead
    std::cerr << "course: " << e.what() << std::endl;
    return 1;
}
