#include "plane.hpp"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is loaded on demand (rccl_api below)

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <stdexcept>

#include "vtk_io.hpp"

namespace {
constexpr int kTileRows = 16;  // one workgroup row of 8x8 wavefront tiles
constexpr std::size_t kProbeEvery = 64;   // blocks layout: every so many frames the walk counts segments per row
constexpr double kRowBaseCost = 3.0;      // per pixel, in segments: what a pixel costs whether or not it meets the grid
                                          // (bench.py --row-base-cost, scripts/sim_scaling.py: the same figure)
constexpr int kBlockQuantum = 8;           // rows: blocks are cut at multiples of the walk's tile height (row_blocks.hpp)
constexpr double kRebalanceGain = 1.08;   // blocks move when the dearest block is this much above what new blocks promise

void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// librccl.so is half a gigabyte of code objects that the HIP runtime would register at start-up of every
// single-GPU run: it is opened only when --exchange rccl asks for it.
struct rccl_api {
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
    bool ok = false;
    static rccl_api& get() {
        static rccl_api api = [] {
            rccl_api a;
            void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
            if (!h) {
                a.error = std::string("cannot load librccl: ") + dlerror();
                return a;
            }
            auto sym = [&](const char* name) {
                void* p = dlsym(h, name);
                if (!p && a.error.empty()) a.error = std::string("librccl lacks ") + name;
                return p;
            };
            a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(sym("ncclCommInitAll"));
            a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
            a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
            a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
            a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
            a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
            a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
            a.ok = a.error.empty();
            return a;
        }();
        return api;
    }
};
}  // namespace

// ------------------------------------------------------------------------------------------------
// pinned images
// ------------------------------------------------------------------------------------------------
image_pool::~image_pool() {
    for (float* p : _free) (void)hipHostFree(p);
}

std::shared_ptr<float> image_pool::take() {
    float* p = nullptr;
    {
        std::lock_guard<std::mutex> hold(_lock);
        if (!_free.empty()) {
            p = _free.back();
            _free.pop_back();
        }
    }
    if (!p)  // portable: visible as pinned memory to every device's context, not only the one current here
        hip_check(hipHostMalloc(reinterpret_cast<void**>(&p), _bytes ? _bytes : 1, hipHostMallocPortable), "hipHostMalloc of an image");
    std::weak_ptr<image_pool> home = shared_from_this();
    return std::shared_ptr<float>(p, [home](float* q) {
        if (auto pool = home.lock()) {
            std::lock_guard<std::mutex> hold(pool->_lock);
            pool->_free.push_back(q);
        } else {
            (void)hipHostFree(q);
        }
    });
}

void object2d::export_to_png(const std::string& filename) const {
    const render_config& cfg = app::instance().config;
    double lo = cfg.png_lo, hi = cfg.png_hi;
    if (!cfg.png_fixed_range) colour_range(_pixels.get(), static_cast<int>(_res_x), static_cast<int>(_res_y), cfg.png_channel, &lo, &hi);
    write_png(filename, _pixels.get(), static_cast<int>(_res_x), static_cast<int>(_res_y), cfg.png_channel, lo, hi);
}

void object2d::export_to_vti(const std::string& filename) const {
    write_vti(filename, _pixels.get(), static_cast<int>(_res_x), static_cast<int>(_res_y),
              !app::instance().config.raw_vti);
}

std::vector<int> parse_device_list(const std::string& text) {
    std::vector<int> out;
    std::size_t pos = 0;
    while (pos <= text.size()) {
        const std::size_t comma = std::min(text.find(',', pos), text.size());
        const std::string part = text.substr(pos, comma - pos);
        if (part.empty()) throw std::runtime_error("bad device list '" + text + "'");
        const std::size_t dash = part.find('-');
        try {
            if (dash == std::string::npos) {
                out.push_back(std::stoi(part));
            } else {
                const int a = std::stoi(part.substr(0, dash)), b = std::stoi(part.substr(dash + 1));
                if (b < a) throw std::runtime_error("descending range");
                for (int d = a; d <= b; ++d) out.push_back(d);
            }
        } catch (const std::exception&) {
            throw std::runtime_error("bad device list '" + text + "'");
        }
        pos = comma + 1;
    }
    if (out.empty()) throw std::runtime_error("empty device list");
    return out;
}

// ------------------------------------------------------------------------------------------------
// device-side exchange (rccl / p2p): per-device strips, the root's full image, communicators
// ------------------------------------------------------------------------------------------------
struct multi_gpu {
    std::vector<int> devices;
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> done;
    std::vector<float*> strip;
    std::vector<int> n_rows;   // local rows per device
    float* root_frame = nullptr;
    std::vector<ncclComm_t> comm;
    bool rccl_ready = false;
    std::string rccl_note;

    ~multi_gpu() {
        for (std::size_t r = 0; r < devices.size(); ++r) {
            (void)hipSetDevice(devices[r]);
            if (r < comm.size() && comm[r]) (void)rccl_api::get().CommDestroy(comm[r]);
            if (r < strip.size() && strip[r]) (void)hipFree(strip[r]);
            if (r < done.size() && done[r]) (void)hipEventDestroy(done[r]);
            if (r < stream.size() && stream[r]) (void)hipStreamDestroy(stream[r]);
        }
        if (root_frame) {
            (void)hipSetDevice(devices[0]);
            (void)hipFree(root_frame);
        }
    }
};

std::string rccl_selftest(int device) {
    rccl_api& nccl = rccl_api::get();
    if (!nccl.ok) throw std::runtime_error(nccl.error);
    hip_check(hipSetDevice(device), "hipSetDevice");
    ncclComm_t comm = nullptr;
    int devs[1] = {device};
    ncclResult_t rc = nccl.CommInitAll(&comm, 1, devs);
    if (rc != ncclSuccess) throw std::runtime_error(std::string("ncclCommInitAll: ") + nccl.GetErrorString(rc));
    // a "strip" of 5 tiles (the last one short) scattered to every other tile of a "frame", as rank r's tiles are
    // scattered into the root's image
    const std::size_t row_floats = 2 * 301, tile_floats = row_floats * kTileRows;
    const int n_tiles = 5, last_rows = 7;
    const std::size_t strip_floats = tile_floats * (n_tiles - 1) + row_floats * last_rows, frame_floats = tile_floats * 2 * n_tiles;
    std::vector<float> host(strip_floats);
    for (std::size_t k = 0; k < strip_floats; ++k) host[k] = static_cast<float>(k % 100003) + 0.25f;
    float *strip = nullptr, *frame = nullptr;
    hipStream_t s = nullptr;
    hip_check(hipMalloc(reinterpret_cast<void**>(&strip), strip_floats * sizeof(float)), "hipMalloc");
    hip_check(hipMalloc(reinterpret_cast<void**>(&frame), frame_floats * sizeof(float)), "hipMalloc");
    hip_check(hipStreamCreate(&s), "hipStreamCreate");
    hip_check(hipMemcpy(strip, host.data(), strip_floats * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy");
    hip_check(hipMemset(frame, 0, frame_floats * sizeof(float)), "hipMemset");
    rc = nccl.GroupStart();
    for (int lt = 0; lt < n_tiles && rc == ncclSuccess; ++lt) {
        const std::size_t count = (lt + 1 == n_tiles ? static_cast<std::size_t>(last_rows) : static_cast<std::size_t>(kTileRows)) * row_floats;
        rc = nccl.Send(strip + tile_floats * lt, count, ncclFloat, 0, comm, s);
        if (rc == ncclSuccess) rc = nccl.Recv(frame + tile_floats * (2 * lt + 1), count, ncclFloat, 0, comm, s);
    }
    const ncclResult_t end = nccl.GroupEnd();
    if (rc != ncclSuccess || end != ncclSuccess)
        throw std::runtime_error(std::string("RCCL self-test: ") + nccl.GetErrorString(rc != ncclSuccess ? rc : end));
    hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
    std::vector<float> back(frame_floats);
    hip_check(hipMemcpy(back.data(), frame, frame_floats * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy");
    std::size_t wrong = 0;
    for (int gt = 0; gt < 2 * n_tiles; ++gt) {
        const int lt = gt / 2;
        const std::size_t rows = (gt % 2 == 0) ? 0 : (lt + 1 == n_tiles ? last_rows : kTileRows);
        for (std::size_t k = 0; k < tile_floats; ++k) {
            const float want = k < rows * row_floats ? host[tile_floats * static_cast<std::size_t>(lt) + k] : 0.0f;
            wrong += back[tile_floats * static_cast<std::size_t>(gt) + k] != want;
        }
    }
    // blocks layout: the frame is three blocks of rows; block 0 is "rendered in place" by the root, blocks 1 and 2 are
    // peers' strips, each landed by ONE ncclSend / ncclRecv pair (one message per GPU and frame), both
    // pairs in one group — the root receives from all its peers at once
    const std::size_t rows_total = 2 * static_cast<std::size_t>(n_tiles) * kTileRows;
    const std::size_t b_begin[3] = {0, 37, 37 + 64}, b_rows[3] = {37, 64, rows_total - 37 - 64};
    std::size_t wrong_blocks = 0;
    {
        hip_check(hipMemset(frame, 0, frame_floats * sizeof(float)), "hipMemset");
        float* peer[3] = {nullptr, nullptr, nullptr};
        std::vector<std::vector<float>> src(3);
        for (int b = 1; b < 3; ++b) {
            src[b].resize(b_rows[b] * row_floats);
            for (std::size_t k = 0; k < src[b].size(); ++k) src[b][k] = static_cast<float>((k * 7 + static_cast<std::size_t>(b)) % 99991) + 0.5f;
            hip_check(hipMalloc(reinterpret_cast<void**>(&peer[b]), src[b].size() * sizeof(float)), "hipMalloc");
            hip_check(hipMemcpy(peer[b], src[b].data(), src[b].size() * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy");
        }
        rc = nccl.GroupStart();
        for (int b = 1; b < 3 && rc == ncclSuccess; ++b) {
            rc = nccl.Send(peer[b], b_rows[b] * row_floats, ncclFloat, 0, comm, s);
            if (rc == ncclSuccess) rc = nccl.Recv(frame + b_begin[b] * row_floats, b_rows[b] * row_floats, ncclFloat, 0, comm, s);
        }
        const ncclResult_t end2 = nccl.GroupEnd();
        if (rc != ncclSuccess || end2 != ncclSuccess)
            throw std::runtime_error(std::string("RCCL self-test (blocks): ") + nccl.GetErrorString(rc != ncclSuccess ? rc : end2));
        hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
        hip_check(hipMemcpy(back.data(), frame, frame_floats * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy");
        for (std::size_t k = 0; k < b_rows[0] * row_floats; ++k) wrong_blocks += back[k] != 0.0f;
        for (int b = 1; b < 3; ++b)
            for (std::size_t k = 0; k < src[b].size(); ++k) wrong_blocks += back[b_begin[b] * row_floats + k] != src[b][k];
        for (int b = 1; b < 3; ++b) (void)hipFree(peer[b]);
    }
    (void)nccl.CommDestroy(comm);
    (void)hipStreamDestroy(s);
    (void)hipFree(strip);
    (void)hipFree(frame);
    if (wrong) throw std::runtime_error("RCCL self-test: " + std::to_string(wrong) + " floats in the wrong place");
    if (wrong_blocks) throw std::runtime_error("RCCL self-test (blocks): " + std::to_string(wrong_blocks) + " floats in the wrong place");
    return "RCCL self-test ok: librccl loaded, communicator over device " + std::to_string(device) + ", " + std::to_string(n_tiles) +
           " grouped ncclSend/ncclRecv pairs landed " + std::to_string(strip_floats) + " floats at their tile offsets; 2 pairs (one per peer) landed " +
           std::to_string((b_rows[1] + b_rows[2]) * row_floats) + " floats as whole row blocks";
}

// ------------------------------------------------------------------------------------------------
// automatic boundaries (plane.cpp:278-288 over object3d_base::get_boundaries, object3d_base.cpp:221-255, over
// tetra::get_boundaries, tetra.cpp:18-42): the x / y bounding box of every object's TRANSFORMED vertices.  The
// vertices are transformed on the GPU each frame; for this one-off the host applies the same rotation lists with
// the reference's arithmetic (tetra.cpp:44-62).  Never used by the reference's own main (main.cpp:83 passes the
// domain), so this path is not on the hot path either.
// ------------------------------------------------------------------------------------------------
std::array<double, 4> bounding_box(std::vector<object3d_base>& objects3d) {
    bool any = false;
    double x_max = 0, x_min = 0, y_max = 0, y_min = 0;
    for (object3d_base& obj : objects3d) {
        const object3d_data& d = *obj.get_pointer();
        const std::vector<double>& pts = d.kind == tetra_type::solid ? d.soup : d.points;
        const std::size_t n = pts.size() / 3;
        double bx_max = -HUGE_VAL, bx_min = HUGE_VAL, by_max = -HUGE_VAL, by_min = HUGE_VAL;
#pragma omp parallel for schedule(static) reduction(max : bx_max, by_max) reduction(min : bx_min, by_min)
        for (std::size_t i = 0; i < n; ++i) {
            double p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
            for (const c5_rotation& r : d.rotations) {
                if (r.axis == 0) {  // tetra.cpp:44-48
                    const double y0 = p[1];
                    p[1] = p[1] * cos(r.angle) - p[2] * sin(r.angle);
                    p[2] = y0 * sin(r.angle) + p[2] * cos(r.angle);
                } else {            // tetra.cpp:51-62
                    p[0] -= r.x0;
                    const double xs = p[0];
                    p[0] = p[0] * cos(r.angle) - p[2] * sin(r.angle);
                    p[2] = xs * sin(r.angle) + p[2] * cos(r.angle);
                    p[0] += r.x0;
                }
            }
            bx_max = std::max(bx_max, p[0]);
            bx_min = std::min(bx_min, p[0]);
            by_max = std::max(by_max, p[1]);
            by_min = std::min(by_min, p[1]);
        }
        if (n == 0) continue;
        x_max = any ? std::max(x_max, bx_max) : bx_max;
        x_min = any ? std::min(x_min, bx_min) : bx_min;
        y_max = any ? std::max(y_max, by_max) : by_max;
        y_min = any ? std::min(y_min, by_min) : by_min;
        any = true;
    }
    if (!any) throw std::runtime_error("plane initializer. empty set of objects to render");
    return {x_max, x_min, y_max, y_min};
}

// ------------------------------------------------------------------------------------------------
// plane
// ------------------------------------------------------------------------------------------------
void plane::check(int rc, const char* what, std::size_t dev) {
    if (rc == C5_OK) return;
    // the reference throws std::runtime_error from the same places (plane.cpp:40,152,263,270)
    throw std::runtime_error(std::string(what) + ": " + c5_last_error(dev < _ctx.size() ? _ctx[dev] : nullptr));
}

plane::plane(std::size_t res_x, std::size_t res_y, std::vector<object3d_base> objects3d,
             std::vector<double> global_boundaries, std::vector<int> devices, exchange_mode exchange, row_layout layout)
    : _devices(std::move(devices)), _exchange(exchange), _layout(layout) {
    if (!global_boundaries.empty() && global_boundaries.size() != 4)
        throw std::runtime_error("plane initializer. wrong manual boundaries");  // plane.cpp:262-264
    if (objects3d.empty())
        throw std::runtime_error("plane initializer. empty set of objects to render");  // plane.cpp:269-271
    if (global_boundaries.empty()) {
        const std::array<double, 4> b = bounding_box(objects3d);  // plane.cpp:278-288
        global_boundaries.assign(b.begin(), b.end());
    }
    if (_devices.empty()) throw std::runtime_error("plane initializer. no GPU given");
    _x = res_x;
    _y = res_y;
    const int world = static_cast<int>(_devices.size());
    if (world > 1 && res_y < static_cast<std::size_t>(world)) throw std::runtime_error("plane initializer. more GPUs than image rows");
    _retry_seen.assign(_devices.size(), 0);

    // volume grids are merged into one indexed grid (the reference concatenates tetra vectors,
    // plane.cpp:290-293); solids keep one slot each, in order
    std::vector<double> pts, a, q;
    std::vector<int32_t> cells;
    struct solid_ref {
        const object3d_data* d;
        int slot;
    };
    std::vector<solid_ref> solids;
    int next_slot = 0;
    for (object3d_base& obj : objects3d) {
        const object3d_data& d = *obj.get_pointer();
        if (d.kind == tetra_type::solid) {
            if (next_slot >= C5_MAX_SOLIDS) throw std::runtime_error("too many solid objects");
            solids.push_back({&d, next_slot});
            _slot_of_object.push_back(next_slot++);
        } else {
            const int32_t base = static_cast<int32_t>(pts.size() / 3);
            pts.insert(pts.end(), d.points.begin(), d.points.end());
            for (int32_t id : d.cells) cells.push_back(id + base);
            a.insert(a.end(), d.value0.begin(), d.value0.end());
            q.insert(q.end(), d.value1.begin(), d.value1.end());
            _slot_of_object.push_back(-1);
        }
    }
    _views.solids.resize(static_cast<std::size_t>(next_slot));

    // blocks layout: equal blocks to start with; the first frame counts its segments per row, and the blocks are
    // balanced by that before the second (read_row_costs / apply_blocks)
    if (world > 1 && _layout == row_layout::blocks)
        for (int r = 0; r < world; ++r) {
            const int lo = static_cast<int>(res_y * static_cast<std::size_t>(r) / static_cast<std::size_t>(world));
            const int hi = static_cast<int>(res_y * static_cast<std::size_t>(r + 1) / static_cast<std::size_t>(world));
            _blocks.emplace_back(lo, hi - lo);
        }

    // one context per device, the grid replicated (52 MB at 1M cells; pixels are independent,
    // plane.cpp:161-169, so nothing but the image is ever exchanged)
    for (int r = 0; r < world; ++r) {
        c5_context* ctx = nullptr;
        if (c5_create(_devices[static_cast<std::size_t>(r)], &ctx) != C5_OK)
            throw std::runtime_error(std::string("c5_create: ") + c5_last_error(nullptr));
        _ctx.push_back(ctx);
        const std::size_t k = static_cast<std::size_t>(r);
        for (const solid_ref& s : solids)
            check(c5_set_solid(ctx, s.slot, s.d->soup.data(), static_cast<int64_t>(s.d->soup.size() / 12), s.d->colour),
                  "c5_set_solid", k);
        if (!cells.empty())
            check(c5_upload_grid(ctx, pts.data(), static_cast<int64_t>(pts.size() / 3), cells.data(),
                                 static_cast<int64_t>(cells.size() / 4), a.data(), q.data()),
                  "c5_upload_grid", k);
        if (world > 1 && _layout == row_layout::tiles) check(c5_set_row_tiles(ctx, kTileRows, r, world), "c5_set_row_tiles", k);
        if (world > 1 && _layout == row_layout::blocks) check(c5_set_row_range(ctx, _blocks[k].first, _blocks[k].second), "c5_set_row_range", k);
        check(c5_set_image(ctx, static_cast<int>(res_x), static_cast<int>(res_y), global_boundaries.data()), "c5_set_image", k);
        check(c5_set_alpha_limit(ctx, app::instance().config.limit_alpha_value), "c5_set_alpha_limit", k);  // line.cpp:204
        if (app::instance().config.reference_algorithm) check(c5_set_option(ctx, "algorithm", 1.0), "c5_set_option", k);
        // --bench times frames as if each were a new one: no per-view data carried from frame to frame (a real sweep with
        // a fixed view - the donor sweep -D, main.cpp:112-116 - keeps them: the library's "view_cache")
        if (app::instance().config.bench > 0) check(c5_set_option(ctx, "view_cache", 0.0), "c5_set_option", k);
        // test hook: start from an overflow pool of this many records, so that the C5_RETRY handling of trace_rays
        // (frames in flight rendered again, in order, each with its own views) is exercised end to end
        if (const char* starve = std::getenv("C5_TEST_ENTRY_POOL"))
            check(c5_set_option(ctx, "entry_pool", std::atof(starve)), "c5_set_option", k);
    }
    update_views(objects3d);
    _pool = std::make_shared<image_pool>(res_x * res_y * 2 * sizeof(float));

    if (_exchange != exchange_mode::host) {
        _mg = std::make_unique<multi_gpu>();
        multi_gpu& m = *_mg;
        m.devices = _devices;
        m.stream.assign(_devices.size(), nullptr);
        m.done.assign(_devices.size(), nullptr);
        m.strip.assign(_devices.size(), nullptr);
        m.n_rows.assign(_devices.size(), 0);
        for (std::size_t r = 0; r < _devices.size(); ++r) {
            hip_check(hipSetDevice(_devices[r]), "hipSetDevice");
            hip_check(hipStreamCreateWithFlags(&m.stream[r], hipStreamNonBlocking), "hipStreamCreate");
            hip_check(hipEventCreateWithFlags(&m.done[r], hipEventDisableTiming), "hipEventCreate");
            check(c5_set_stream(_ctx[r], m.stream[r]), "c5_set_stream", r);
            check(c5_local_rows(_ctx[r], &m.n_rows[r]), "c5_local_rows", r);
            // blocks move between frames: a strip has room for any block (the whole image: 35 MB at 2400x1800)
            const std::size_t rows = _layout == row_layout::blocks ? res_y : static_cast<std::size_t>(m.n_rows[r]);
            const std::size_t bytes = rows * res_x * 2 * sizeof(float);
            // (the root's block is rendered straight into the root image: it needs no strip in the blocks layout)
            if (!(r == 0 && world > 1 && _layout == row_layout::blocks))
                hip_check(hipMalloc(reinterpret_cast<void**>(&m.strip[r]), bytes ? bytes : 8), "hipMalloc of a strip");
            if (r > 0 && _devices[r] != _devices[0]) {
                const hipError_t e = hipDeviceEnablePeerAccess(_devices[0], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) hip_check(e, "hipDeviceEnablePeerAccess");
                (void)hipGetLastError();
            }
        }
        hip_check(hipSetDevice(_devices[0]), "hipSetDevice");
        hip_check(hipMalloc(reinterpret_cast<void**>(&m.root_frame), res_x * res_y * 2 * sizeof(float)), "hipMalloc of the root image");
        if (_exchange == exchange_mode::rccl && world > 1) {
            rccl_api& nccl = rccl_api::get();
            if (!nccl.ok) {
                m.rccl_note = nccl.error + " - exchanging by peer copies instead";
            } else {
                m.comm.assign(_devices.size(), nullptr);
                const ncclResult_t rc = nccl.CommInitAll(m.comm.data(), world, _devices.data());
                if (rc == ncclSuccess) {
                    m.rccl_ready = true;
                } else {
                    // e.g. the same GPU named twice (a rehearsal on one GPU): RCCL refuses duplicate devices
                    m.comm.clear();
                    m.rccl_note = std::string("ncclCommInitAll: ") + nccl.GetErrorString(rc) + " - exchanging by peer copies instead";
                }
            }
            if (!m.rccl_ready) std::cerr << "course: " << m.rccl_note << std::endl;
        }
    }
}

plane::~plane() {
    for (std::size_t r = 0; r < _ctx.size(); ++r) {
        if (!_ctx[r]) continue;
        while (_exchange == exchange_mode::host && !_flight.empty()) {  // nothing may still be copying into a pooled image
            for (c5_context* c : _ctx) (void)c5_render_host_wait(c);
            _flight.pop_front();
        }
        (void)c5_synchronize(_ctx[r]);
    }
    _mg.reset();
    for (c5_context* c : _ctx)
        if (c) c5_destroy(c);
}

void plane::send_views(const views_t& v) {
    for (std::size_t r = 0; r < _ctx.size(); ++r) {
        check(c5_set_view(_ctx[r], v.grid.data(), static_cast<int>(v.grid.size())), "c5_set_view", r);
        for (std::size_t s = 0; s < v.solids.size(); ++s)
            check(c5_set_solid_view(_ctx[r], static_cast<int>(s), v.solids[s].data(), static_cast<int>(v.solids[s].size())),
                  "c5_set_solid_view", r);
    }
}

plane::views_t plane::views_of(std::vector<object3d_base>& objects3d) const {
    views_t v;
    v.solids.resize(_views.solids.size());
    bool grid_view_set = false;
    for (std::size_t k = 0; k < objects3d.size() && k < _slot_of_object.size(); ++k) {
        const object3d_data& d = *objects3d[k].get_pointer();
        if (_slot_of_object[k] < 0) {
            if (!grid_view_set) v.grid = d.rotations;
            grid_view_set = true;
        } else {
            v.solids[static_cast<std::size_t>(_slot_of_object[k])] = d.rotations;
        }
    }
    return v;
}

void plane::set_views(const views_t& v) {
    _views = v;
    send_views(_views);
}

void plane::start(frame_t& f) {
    const bool blocks = _ctx.size() > 1 && _layout == row_layout::blocks;
    if (_stage_on.size() != _ctx.size()) _stage_on.assign(_ctx.size(), -1);
    for (std::size_t r = 0; r < _ctx.size(); ++r) {
        if (blocks) check(c5_set_option(_ctx[r], "row_costs", f.probe ? 1.0 : 0.0), "c5_set_option", r);
        // (a probe frame, and the frames issued while one is in flight: read_row_costs reads the time of the LAST frame each
        // device has completed)
        bool probing = f.probe;
        for (const frame_t& g : _flight) probing = probing || g.probe;
        const int want_times = (_stage_times || probing) ? 1 : 0;
        if (_stage_on[r] != want_times) {
            check(c5_set_option(_ctx[r], "stage_timing", want_times), "c5_set_option", r);
            _stage_on[r] = want_times;
        }
        if (_exchange == exchange_mode::host) {
            check(c5_render_frame_rows_async(_ctx[r], f.image.get()), "find_intersections", r);
        } else {
            // blocks: the root's own block is rendered in place, at its final offset of the root image
            float* const target = (blocks && r == 0) ? _mg->root_frame + static_cast<std::size_t>(_blocks[0].first) * _x * 2 : _mg->strip[r];
            check(c5_render_device(_ctx[r], target), "find_intersections", r);
        }
    }
}

bool plane::retry_seen() {
    bool any = false;
    for (char& c : _retry_seen) {
        any = any || c != 0;
        c = 0;
    }
    return any;
}

void plane::find_intersections() {
    const std::size_t limit = _exchange == exchange_mode::host ? C5_HOST_RING : 1;
    if (_flight.size() >= limit)
        throw std::runtime_error("find_intersections: " + std::to_string(limit) + " frame(s) already in flight, call trace_rays first");
    if (_rebalance_due) {
        // The rows are laid out anew between frames: whatever is in flight finishes first and is parked, in order,
        // for the trace_rays calls to come.  (A probe frame among them is not read: its layout is about to go.)
        while (!_flight.empty()) {
            complete_front();
            _parked.push_back(std::move(_flight.front()));
            _flight.pop_front();
        }
        apply_blocks(_wanted_blocks);
        _rebalance_due = false;
    }
    frame_t f;
    f.image = _pool->take();
    f.views = _views;
    f.probe = _ctx.size() > 1 && _layout == row_layout::blocks && (_issued % kProbeEvery == 0);
    ++_issued;
    start(f);
    _flight.push_back(std::move(f));
}

// blocks layout, after a probe frame completed: every device hands over the segments of its rows; together they are
// the cost of every row of the image.  If balanced blocks would make the dearest block at least kRebalanceGain
// cheaper than it is now, the next find_intersections lays the rows out anew.
void plane::read_row_costs() {
    _row_cost.assign(_y, 0);
    bool complete = true;
    for (std::size_t r = 0; r < _ctx.size(); ++r) {
        const int n = _blocks[r].second;
        if (n <= 0) continue;
        const int rc = c5_get_row_costs(_ctx[r], _row_cost.data() + _blocks[r].first, n);
        if (rc == C5_RETRY) {  // a frame in flight behind the probe overflowed: remembered for its trace_rays; no costs this time
            _retry_seen[r] = 1;
            complete = false;
        } else {
            check(rc, "c5_get_row_costs", r);
        }
    }
    if (!complete) return;  // the next probe frame decides
    const double base = kRowBaseCost * static_cast<double>(_x);
    // What the devices TOOK for their blocks (GPU time of the last frame each has completed: the probe frame or one behind
    // it, same layout) corrects the model block by block: a segment costs more in the tiles of an oblique face than in the
    // grid's body, and a share costs something whatever its rows (row_blocks.hpp: time_weighted_row_costs).
    std::vector<double> times(_ctx.size(), 0.0);
    bool timed = true;
    for (std::size_t r = 0; r < _ctx.size(); ++r) {
        c5_stats st{};
        int rc = c5_get_stats(_ctx[r], &st);
        if (rc == C5_RETRY) {  // (settled by this very call: remembered for the frames in flight, as in stats())
            _retry_seen[r] = 1;
            rc = C5_OK;
        }
        check(rc, "c5_get_stats", r);
        times[r] = static_cast<double>(st.ms_total);
        if (!(times[r] > 0.0) && _blocks[r].second > 0) timed = false;
    }
    std::vector<double> cost(_row_cost.size());
    if (timed) {
        cost = time_weighted_row_costs(_row_cost, base, _blocks, times);
    } else {
        for (std::size_t k = 0; k < cost.size(); ++k) cost[k] = static_cast<double>(_row_cost[k]) + base;
    }
    const std::vector<std::pair<int, int>> want = balanced_row_blocks(cost, static_cast<int>(_ctx.size()), kBlockQuantum);
    auto dearest = [&](const std::vector<std::pair<int, int>>& blocks) {
        double top = 0.0;
        for (const auto& b : blocks) {
            double c = 0.0;
            for (int k = 0; k < b.second; ++k) c += cost[static_cast<std::size_t>(b.first + k)];
            top = std::max(top, c);
        }
        return top;
    };
    if (want != _blocks && dearest(_blocks) > kRebalanceGain * dearest(want)) {
        _wanted_blocks = want;
        _rebalance_due = true;
    }
}

void plane::apply_blocks(const std::vector<std::pair<int, int>>& blocks) {
    for (std::size_t r = 0; r < _ctx.size(); ++r) {
        int rc = c5_set_row_range(_ctx[r], blocks[r].first, blocks[r].second);
        if (rc == C5_RETRY) rc = C5_OK;  // nothing is in flight here: a retry belongs to frames already completed again
        check(rc, "c5_set_row_range", r);
        if (_mg) _mg->n_rows[r] = blocks[r].second;
    }
    _blocks = blocks;
    ++_rebalances;
}

// rccl / p2p: every strip is known complete (status checked) before anything is exchanged; every block (tile) lands
// at its final offset of the root GPU's image; one copy from there to the host.
void plane::finish_exchange(frame_t& f) {
    multi_gpu& m = *_mg;
    const int world = static_cast<int>(_ctx.size());
    const bool blocks = world > 1 && _layout == row_layout::blocks;
    const std::size_t row_floats = _x * 2, tile_floats = row_floats * kTileRows;
    const std::size_t row_bytes = row_floats * sizeof(float), tile_bytes = tile_floats * sizeof(float);
    auto target_of = [&](std::size_t r) { return (blocks && r == 0) ? m.root_frame + static_cast<std::size_t>(_blocks[0].first) * row_floats : m.strip[r]; };
    for (std::size_t r = 0; r < _ctx.size(); ++r) {
        int rc = c5_synchronize(_ctx[r]);
        if (_retry_seen[r]) {  // stats() ran into the retry first (and with it cleared the library's own note of it)
            _retry_seen[r] = 0;
            if (rc == C5_OK) rc = C5_RETRY;
        }
        for (int attempt = 0; rc == C5_RETRY && attempt < 3; ++attempt) {  // an internal buffer grew: this device renders again
            ++_retries;
            send_views(f.views);
            check(c5_render_device(_ctx[r], target_of(r)), "trace_rays", r);
            rc = c5_synchronize(_ctx[r]);
            send_views(_views);
        }
        check(rc, "trace_rays", r);
    }
    const int n_tiles = static_cast<int>((_y + kTileRows - 1) / kTileRows);
    auto rows_of_tile = [&](int gt) { return std::min<std::size_t>(kTileRows, _y - static_cast<std::size_t>(gt) * kTileRows); };
    auto copy_own_tiles = [&](std::size_t r, hipStream_t s) {  // 2-D copy: one "row" of it = one tile
        const int whole = m.n_rows[r] / kTileRows, rest = m.n_rows[r] - whole * kTileRows;
        float* const first = m.root_frame + tile_floats * r;
        if (world == 1) {
            hip_check(hipMemcpyAsync(m.root_frame, m.strip[r], row_bytes * m.n_rows[r], hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
            return;
        }
        if (whole > 0)
            hip_check(hipMemcpy2DAsync(first, tile_bytes * world, m.strip[r], tile_bytes, tile_bytes, static_cast<std::size_t>(whole),
                                       hipMemcpyDeviceToDevice, s), "hipMemcpy2DAsync");
        if (rest > 0)
            hip_check(hipMemcpyAsync(first + tile_floats * world * whole, m.strip[r] + tile_floats * whole, row_bytes * rest,
                                     hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
    };
    if (_exchange == exchange_mode::rccl && m.rccl_ready) {
        rccl_api& nccl = rccl_api::get();
        ncclResult_t rc = nccl.GroupStart();
        if (blocks) {
            // ONE message per GPU and frame: peer r's block, contiguous in the image, received at its
            // final offset of the root's image; the root receives from all its peers at once (7 xGMI links)
            for (int r = 1; r < world && rc == ncclSuccess; ++r) {
                const std::size_t count = static_cast<std::size_t>(_blocks[static_cast<std::size_t>(r)].second) * row_floats;
                if (count == 0) continue;
                rc = nccl.Send(m.strip[static_cast<std::size_t>(r)], count, ncclFloat, 0, m.comm[static_cast<std::size_t>(r)], m.stream[static_cast<std::size_t>(r)]);
                if (rc == ncclSuccess)
                    rc = nccl.Recv(m.root_frame + static_cast<std::size_t>(_blocks[static_cast<std::size_t>(r)].first) * row_floats, count, ncclFloat, r,
                                   m.comm[0], m.stream[0]);
            }
        } else {
            for (int gt = 0; gt < n_tiles && rc == ncclSuccess; ++gt) {
                const int r = gt % world, lt = gt / world;
                if (r == 0) continue;
                const std::size_t count = rows_of_tile(gt) * row_floats;
                rc = nccl.Send(m.strip[static_cast<std::size_t>(r)] + tile_floats * lt, count, ncclFloat, 0,
                              m.comm[static_cast<std::size_t>(r)], m.stream[static_cast<std::size_t>(r)]);
                if (rc == ncclSuccess)
                    rc = nccl.Recv(m.root_frame + tile_floats * gt, count, ncclFloat, r, m.comm[0], m.stream[0]);
            }
        }
        const ncclResult_t end = nccl.GroupEnd();
        if (rc != ncclSuccess || end != ncclSuccess)
            throw std::runtime_error(std::string("RCCL exchange failed: ") + nccl.GetErrorString(rc != ncclSuccess ? rc : end));
        hip_check(hipSetDevice(_devices[0]), "hipSetDevice");
        if (!blocks) copy_own_tiles(0, m.stream[0]);
    } else {
        for (std::size_t r = 0; r < _ctx.size(); ++r) {
            hip_check(hipSetDevice(_devices[r]), "hipSetDevice");
            if (blocks) {
                if (r == 0) continue;  // rendered in place
                // one peer copy per GPU, on the sending device's stream, into the root's memory
                hip_check(hipMemcpyAsync(m.root_frame + static_cast<std::size_t>(_blocks[r].first) * row_floats, m.strip[r],
                                         row_bytes * static_cast<std::size_t>(_blocks[r].second), hipMemcpyDeviceToDevice, m.stream[r]), "hipMemcpyAsync (peer)");
            } else {
                copy_own_tiles(r, m.stream[r]);  // runs on the sending device's stream, writes the root's memory
            }
            if (r > 0) hip_check(hipEventRecord(m.done[r], m.stream[r]), "hipEventRecord");
        }
        hip_check(hipSetDevice(_devices[0]), "hipSetDevice");
        for (std::size_t r = 1; r < _ctx.size(); ++r) hip_check(hipStreamWaitEvent(m.stream[0], m.done[r], 0), "hipStreamWaitEvent");
    }
    hip_check(hipMemcpyAsync(f.image.get(), m.root_frame, _x * _y * 2 * sizeof(float), hipMemcpyDeviceToHost, m.stream[0]),
              "hipMemcpyAsync to the host");
    hip_check(hipStreamSynchronize(m.stream[0]), "hipStreamSynchronize");
}

// Wait for the oldest frame in flight; on C5_RETRY (an internal buffer was too small: that frame and every frame
// enqueued since are suspect on every device) let them all finish, then render them again, in order, each with its
// own views.
void plane::complete_front() {
    if (_exchange != exchange_mode::host) {
        finish_exchange(_flight.front());
        return;
    }
    for (int attempt = 0;; ++attempt) {
        bool retry = false;
        for (std::size_t r = 0; r < _ctx.size(); ++r) {
            const int rc = c5_render_host_wait(_ctx[r]);
            if (rc == C5_RETRY)
                retry = true;
            else
                check(rc, "trace_rays", r);
        }
        if (retry_seen()) retry = true;  // stats() / the row-cost read met the retry before this wait did
        if (!retry) break;
        if (attempt >= 3) throw std::runtime_error("trace_rays: frames kept being reported incomplete");
        ++_retries;
        for (std::size_t k = 1; k < _flight.size(); ++k)
            for (std::size_t r = 0; r < _ctx.size(); ++r) {
                const int rc = c5_render_host_wait(_ctx[r]);
                if (rc != C5_RETRY) check(rc, "trace_rays", r);
            }
        (void)retry_seen();
        for (frame_t& f : _flight) {
            send_views(f.views);
            start(f);
        }
        send_views(_views);
    }
}

object2d plane::trace_rays(tetra_value value_alpha, tetra_value value_Q) {
    if (value_alpha != tetra_value::alpha || value_Q != tetra_value::Q)
        throw std::runtime_error("trace_rays: only (alpha, Q) is supported");
    if (!_parked.empty()) {  // completed while the rows were laid out anew
        frame_t f = std::move(_parked.front());
        _parked.pop_front();
        return object2d(std::move(f.image), _x, _y);
    }
    if (_flight.empty()) find_intersections();
    complete_front();
    frame_t f = std::move(_flight.front());
    _flight.pop_front();
    if (f.probe) read_row_costs();
    return object2d(std::move(f.image), _x, _y);
}

c5_stats plane::stats() {
    c5_stats sum{};
    for (std::size_t r = 0; r < _ctx.size(); ++r) {
        c5_stats st{};
        int rc = c5_get_stats(_ctx[r], &st);
        if (rc == C5_RETRY) {
            // An internal buffer was too small for a frame since the last look, and by reporting it here the
            // library has settled it (pool grown, failure words cleared).  The counts are still those of the last
            // frame; the frames in flight are incomplete and the next trace_rays renders them again.
            _retry_seen[r] = 1;
            rc = C5_OK;
        }
        check(rc, "c5_get_stats", r);
        sum.segments += st.segments;
        sum.covered_pixels += st.covered_pixels;
        sum.solid_pixels += st.solid_pixels;
        sum.entries += st.entries;
        sum.steps += st.steps;
        sum.pool_entries += st.pool_entries;
        sum.odd_pixels += st.odd_pixels;
        sum.walk_overflow += st.walk_overflow;
        sum.boundary_faces = st.boundary_faces;
        sum.pool_capacity = st.pool_capacity;
        sum.ms_transform = std::max(sum.ms_transform, st.ms_transform);
        sum.ms_records = std::max(sum.ms_records, st.ms_records);
        sum.ms_entries = std::max(sum.ms_entries, st.ms_entries);
        sum.ms_solids = std::max(sum.ms_solids, st.ms_solids);
        sum.ms_walk = std::max(sum.ms_walk, st.ms_walk);
        sum.ms_total = std::max(sum.ms_total, st.ms_total);
    }
    return sum;
}

std::size_t plane::count_all_intersections() { return static_cast<std::size_t>(stats().segments); }
