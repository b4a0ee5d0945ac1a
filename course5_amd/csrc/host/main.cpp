// `course` — command line of the MI355X build.  Same options, banner, timing lines, input and
// output contract as the reference's project/src/main.cpp; the per-pixel OpenMP loops behind
// `plane` are replaced by the HIP kernels in libcourse5_hip.so.
#include <omp.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <exception>
#include <memory>
#include <iostream>
#include <limits>
#include <mutex>
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

// A bounded hand-over between threads: push blocks while `capacity` items wait, pop blocks until an item arrives or
// the queue is closed.
template <class T>
class channel {
public:
    explicit channel(std::size_t capacity) : _capacity(capacity) {}
    void push(T item) {
        std::unique_lock<std::mutex> hold(_lock);
        _room.wait(hold, [&] { return _items.size() < _capacity || _closed; });
        if (_closed) return;
        _items.push_back(std::move(item));
        _ready.notify_one();
    }
    bool pop(T& out) {
        std::unique_lock<std::mutex> hold(_lock);
        _ready.wait(hold, [&] { return !_items.empty() || _closed; });
        if (_items.empty()) return false;
        out = std::move(_items.front());
        _items.pop_front();
        _room.notify_one();
        return true;
    }
    void close() {
        std::lock_guard<std::mutex> hold(_lock);
        _closed = true;
        _ready.notify_all();
        _room.notify_all();
    }

private:
    std::size_t _capacity;
    std::mutex _lock;
    std::condition_variable _ready, _room;
    std::deque<T> _items;
    bool _closed = false;
};

// The files of a sweep are written by a thread of their own (object2d::export_to_vti widens, deflates and encodes
// on the OpenMP team it starts), so that a sweep with output lasts max(render, write) per frame and not their sum:
// the reference's workload is exactly this (utility/rotate_traces.py:16-21 renders 1 500 frames to files).  Frames
// arrive as pinned images of the planes' pools and go back to them when written; the queue bounds how many wait.
class frame_writer {
public:
    explicit frame_writer(int omp_threads) : _queue(4), _thread([this, omp_threads] { run(omp_threads); }) {}
    ~frame_writer() {
        _queue.close();
        if (_thread.joinable()) _thread.join();
    }
    void write(object2d img, std::string name) { _queue.push({std::move(img), std::move(name)}); }
    // all files written; throws what the writer thread ran into
    void finish() {
        _queue.close();
        if (_thread.joinable()) _thread.join();
        if (_error) std::rethrow_exception(_error);
    }
    std::size_t written() const { return _written; }

private:
    struct item {
        object2d img;
        std::string name;
    };
    void run(int omp_threads) {
        omp_set_num_threads(omp_threads);
        item it;
        while (_queue.pop(it)) {
            try {
                if (!_error) {
                    write_frame(it.img, it.name);
                    ++_written;  // (files really written: nothing is counted after a failure)
                }
            } catch (...) {
                if (!_error) _error = std::current_exception();
            }
            it = item{};  // the image goes back to its pool now, not when the next one arrives
        }
    }
    channel<item> _queue;
    std::exception_ptr _error;
    std::size_t _written = 0;
    std::thread _thread;
};

// One thread per plane (per GPU when a sweep is split by frames): it issues the frames dealt to it ahead
// (find_intersections returns at once) and retires them in order, so that N GPUs are driven by N host threads — ten
// HIP calls per 0.7 ms frame and GPU are the whole budget of one.
struct frame_job {
    std::size_t k = 0;
    plane::views_t views;
    bool write = false;
};
class plane_driver {
public:
    // `failed`: shared by the drivers of a sweep and the thread that deals the frames - set by the first driver that throws,
    // so that the others stop issuing and the dealing loop stops dealing (a failing GPU is reported at once, not after the
    // rest of the sweep has been rendered and written)
    plane_driver(plane& p, std::size_t depth, frame_writer* writer, const render_config& config, bool stats_between,
                 std::atomic<bool>& failed)
        : _plane(p), _depth(depth), _writer(writer), _config(config), _stats_between(stats_between), _failed(failed),
          _jobs(2 * depth + 2), _thread([this] { run(); }) {}
    ~plane_driver() {
        _jobs.close();
        if (_thread.joinable()) _thread.join();
    }
    void issue(frame_job job) { _jobs.push(std::move(job)); }
    void finish() {
        _jobs.close();
        if (_thread.joinable()) _thread.join();
        if (_error) std::rethrow_exception(_error);
    }

private:
    void retire() {
        const frame_job job = std::move(_flight.front());
        _flight.pop_front();
        // test hook (C5_TEST_STATS_BETWEEN): the reference's count_all_intersections() may be called between
        // find_intersections() and trace_rays(); a C5_RETRY it runs into must not be lost
        if (_stats_between) (void)_plane.count_all_intersections();
        object2d img = _plane.trace_rays(tetra_value::alpha, tetra_value::Q);
        // (--bench --bench_files: numbered names like a sweep's, whatever --frames says)
        if (job.write && _writer)
            _writer->write(std::move(img), frame_name(_config.destination, job.k, _config.bench > 0 ? std::max<std::size_t>(_config.frames, 2) : _config.frames));
    }
    void run() {
        try {
            frame_job job;
            while (_jobs.pop(job)) {
                if (_failed.load(std::memory_order_relaxed)) continue;  // another driver failed: drain the queue, issue nothing
                while (_flight.size() >= _depth) retire();
                _plane.set_views(job.views);
                _plane.find_intersections();
                _flight.push_back(std::move(job));
            }
            while (!_flight.empty()) retire();
        } catch (...) {
            _error = std::current_exception();
            _failed.store(true, std::memory_order_relaxed);
            frame_job drop;
            while (_jobs.pop(drop)) {}  // keep the dealing thread from blocking on a full queue
        }
    }
    plane& _plane;
    std::size_t _depth;
    frame_writer* _writer;
    const render_config& _config;
    bool _stats_between;
    std::atomic<bool>& _failed;
    channel<frame_job> _jobs;
    std::deque<frame_job> _flight;
    std::exception_ptr _error;
    std::thread _thread;
};
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
    // rows of one frame shared by several GPUs: contiguous cost-balanced blocks need an earlier frame to measure, so
    // a single frame is dealt in cyclic tiles (balanced by construction) unless told otherwise
    const row_layout layout = config.row_layout == "blocks"  ? row_layout::blocks
                              : config.row_layout == "tiles" ? row_layout::tiles
                                                             : (is_sweep ? row_layout::blocks : row_layout::tiles);

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
        planes.push_back(std::make_unique<plane>(config.resolution_x, config.resolution_y, objects, domain, devices, exchange, layout));
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
    if (is_sweep)  // (the frames to come print no stage times: their events, 25 us per frame, are not recorded)
        for (const auto& p : planes) p->stage_times(false);

    // sweep: the grid, its adjacency and the solids stay on the GPU(s); only rotation lists change.  This thread
    // works out the angles of frame k and deals it to the driver of plane k mod N; every driver issues its frames
    // ahead and retires them in order; retired frames go to the writer thread.  While frame k is being written,
    // frames k + 1 ... are rendered and copied to pinned host images.
    const std::size_t per_plane = (by_frames || exchange == exchange_mode::host) ? 2 : 1;
    double* const swept = config.sweep == "X"   ? &view.angle_around_x
                          : config.sweep == "D" ? &view.donor_angle
                          : config.sweep == "I" ? &view.system_initial_angle_around_y
                                                : &view.angle_around_y;
    // Test hook, read from the environment by the product binary on purpose (tests/test_cli_gpu.py drives the real `course`):
    // C5_TEST_STATS_BETWEEN makes every driver call count_all_intersections() between issue and retire, the place the
    // reference allows it; it changes no result.  (C5_TEST_ENTRY_POOL, plane.cpp, starts the entry pool small likewise.)
    const bool stats_between = std::getenv("C5_TEST_STATS_BETWEEN") != nullptr;
    auto run_frames = [&](std::size_t first, std::size_t count, bool advance, bool write) {
        std::unique_ptr<frame_writer> writer;
        if (write) writer = std::make_unique<frame_writer>(static_cast<int>(std::max<std::size_t>(1, config.threads)));
        {
            std::atomic<bool> failed{false};
            std::vector<std::unique_ptr<plane_driver>> drivers;
            for (auto& p : planes) drivers.push_back(std::make_unique<plane_driver>(*p, per_plane, writer.get(), config, stats_between, failed));
            for (std::size_t k = first; k < first + count && !failed.load(std::memory_order_relaxed); ++k) {
                if (advance) *swept += config.sweep_step;
                apply_view(objects[0], roche_lobe ? &objects[1] : nullptr);
                frame_job job;
                job.k = k;
                job.views = planes[k % planes.size()]->views_of(objects);
                job.write = write;
                drivers[k % planes.size()]->issue(std::move(job));
            }
            std::exception_ptr first_error;
            for (auto& d : drivers) {
                try {
                    d->finish();
                } catch (...) {
                    if (!first_error) first_error = std::current_exception();
                }
            }
            if (first_error) std::rethrow_exception(first_error);
        }
        if (writer) writer->finish();
    };
    if (config.bench == 0) {
        if (config.frames > 1) run_frames(1, config.frames - 1, true, true);
        if (config.frames > 1) {
            const auto t3 = timestamp();
            std::cout << config.frames - 1 << " further frames in " << ms_between(t2, t3) << " ms. " << std::endl;
        }
        if (config.print_stats) {
            std::size_t again = 0, moved = 0;
            for (const auto& p : planes) again += p->retries(), moved += p->rebalances();
            std::cout << "Frames rendered again after an internal buffer grew: " << again << std::endl;
            if (devices.size() > 1 && !by_frames) {
                std::cout << "Row blocks laid out anew: " << moved << "; rows per GPU:";
                for (const auto& b : base_plane.row_blocks()) std::cout << " " << b.second;
                std::cout << std::endl;
            }
        }
    } else {
        // --bench: frames rendered and delivered to host memory, no files — or, with --bench_files, written like a
        // sweep's (end to end: render + copy + deflate + file).  One JSON line.
        // --bench_rounds R: the timed run is repeated R times (each `bench` frames, each timed on its own): the line carries the
        // whole run's figure and the fastest and the median round, so that a reader sees the spread and not one draw.
        run_frames(0, config.bench_warmup, false, config.bench_files);  // warm-up through the same path (writer thread, files)
        std::size_t retries0 = 0;
        for (const auto& p : planes) retries0 += p->retries();
        const std::size_t rounds = std::max<std::size_t>(1, config.bench_rounds);
        std::vector<double> round_ms;
        double secs = 0.0;
        for (std::size_t r = 0; r < rounds; ++r) {
            const auto b0 = std::chrono::steady_clock::now();
            run_frames(0, config.bench, config.sweep_step != 0.0, config.bench_files);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - b0).count();
            secs += dt;
            round_ms.push_back(dt * 1e3 / static_cast<double>(config.bench));
        }
        std::sort(round_ms.begin(), round_ms.end());
        const double frames_timed = static_cast<double>(config.bench) * static_cast<double>(rounds);
        std::size_t retries1 = 0, moved = 0;
        for (const auto& p : planes) retries1 += p->retries(), moved += p->rebalances();
        const double rays = static_cast<double>(config.resolution_x) * static_cast<double>(config.resolution_y);
        std::string rows = "[";
        for (const auto& b : base_plane.row_blocks()) rows += (rows.size() > 1 ? ", " : "") + std::to_string(b.second);
        rows += "]";
        std::printf("{\"course_bench\": {\"frames\": %zu, \"rounds\": %zu, \"warmup\": %zu, \"ms_per_frame\": %.4f, \"ms_per_frame_min\": %.4f, "
                    "\"ms_per_frame_median\": %.4f, \"mrays_per_s\": %.1f, \"frames_per_s\": %.2f, "
                    "\"res_x\": %zu, \"res_y\": %zu, \"n_devices\": %zu, \"split\": \"%s\", \"exchange\": \"%s\", \"row_layout\": \"%s\", "
                    "\"rows_per_device\": %s, \"rebalances\": %zu, "
                    "\"sweep\": \"%s\", \"sweep_step\": %g, \"solids\": %s, \"retries\": %zu, "
                    "\"delivered_to\": \"%s\"}}\n",
                    config.bench, rounds, config.bench_warmup, secs * 1e3 / frames_timed, round_ms.front(), round_ms[round_ms.size() / 2],
                    rays * frames_timed / secs / 1e6, frames_timed / secs,
                    config.resolution_x, config.resolution_y, devices.size(),
                    devices.size() == 1 ? "none" : (by_frames ? "frames" : "rows"), by_frames ? "none" : config.exchange.c_str(),
                    (devices.size() == 1 || by_frames) ? "none" : (layout == row_layout::blocks ? "blocks" : "tiles"), rows.c_str(), moved,
                    config.sweep.c_str(), config.sweep_step, config.no_solids ? "false" : "true", retries1 - retries0,
                    config.bench_files ? (config.raw_vti ? "raw .vti files" : "zlib .vti files") : "pinned host memory");
        std::fflush(stdout);
    }
    std::cout << "Result exported. Calculations completed." << std::endl;
    return 0;
} catch (const std::exception& e) {
    // the reference lets std::runtime_error escape and abort; report and fail instead
    std::cerr << "course: " << e.what() << std::endl;
    return 1;
}
