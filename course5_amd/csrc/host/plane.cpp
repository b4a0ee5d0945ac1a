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
    if (!_free.empty()) {
        p = _free.back();
        _free.pop_back();
    } else {
        hip_check(hipHostMalloc(reinterpret_cast<void**>(&p), _bytes ? _bytes : 1, hipHostMallocDefault), "hipHostMalloc of an image");
    }
    std::weak_ptr<image_pool> home = shared_from_this();
    return std::shared_ptr<float>(p, [home](float* q) {
        if (auto pool = home.lock())
            pool->_free.push_back(q);
        else
            (void)hipHostFree(q);
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
    (void)nccl.CommDestroy(comm);
    (void)hipStreamDestroy(s);
    (void)hipFree(strip);
    (void)hipFree(frame);
    if (wrong) throw std::runtime_error("RCCL self-test: " + std::to_string(wrong) + " floats in the wrong place");
    return "RCCL self-test ok: librccl loaded, communicator over device " + std::to_string(device) + ", " + std::to_string(n_tiles) +
           " grouped ncclSend/ncclRecv pairs landed " + std::to_string(strip_floats) + " floats at their tile offsets";
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
             std::vector<double> global_boundaries, std::vector<int> devices, exchange_mode exchange)
    : _devices(std::move(devices)), _exchange(exchange) {
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
        if (world > 1) check(c5_set_row_tiles(ctx, kTileRows, r, world), "c5_set_row_tiles", k);
        check(c5_set_image(ctx, static_cast<int>(res_x), static_cast<int>(res_y), global_boundaries.data()), "c5_set_image", k);
        check(c5_set_alpha_limit(ctx, app::instance().config.limit_alpha_value), "c5_set_alpha_limit", k);  // line.cpp:204
        if (app::instance().config.reference_algorithm) check(c5_set_option(ctx, "algorithm", 1.0), "c5_set_option", k);
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
            const std::size_t bytes = static_cast<std::size_t>(m.n_rows[r]) * res_x * 2 * sizeof(float);
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

void plane::update_views(std::vector<object3d_base>& objects3d) {
    bool grid_view_set = false;
    for (std::size_t k = 0; k < objects3d.size() && k < _slot_of_object.size(); ++k) {
        const object3d_data& d = *objects3d[k].get_pointer();
        if (_slot_of_object[k] < 0) {
            if (!grid_view_set) _views.grid = d.rotations;
            grid_view_set = true;
        } else {
            _views.solids[static_cast<std::size_t>(_slot_of_object[k])] = d.rotations;
        }
    }
    send_views(_views);
}

void plane::start(frame_t& f) {
    for (std::size_t r = 0; r < _ctx.size(); ++r) {
        if (_exchange == exchange_mode::host)
            check(c5_render_frame_rows_async(_ctx[r], f.image.get()), "find_intersections", r);
        else
            check(c5_render_device(_ctx[r], _mg->strip[r]), "find_intersections", r);
    }
}

void plane::find_intersections() {
    const std::size_t limit = _exchange == exchange_mode::host ? C5_HOST_RING : 1;
    if (_flight.size() >= limit)
        throw std::runtime_error("find_intersections: " + std::to_string(limit) + " frame(s) already in flight, call trace_rays first");
    frame_t f;
    f.image = _pool->take();
    f.views = _views;
    start(f);
    _flight.push_back(std::move(f));
}

// rccl / p2p: every strip is known complete (status checked) before anything is exchanged; every tile lands
// at its final offset of the root GPU's image; one copy from there to the host.
void plane::finish_exchange(frame_t& f) {
    multi_gpu& m = *_mg;
    const int world = static_cast<int>(_ctx.size());
    for (std::size_t r = 0; r < _ctx.size(); ++r) {
        int rc = c5_synchronize(_ctx[r]);
        for (int attempt = 0; rc == C5_RETRY && attempt < 3; ++attempt) {  // an internal buffer grew: this device renders again
            ++_retries;
            check(c5_render_device(_ctx[r], m.strip[r]), "trace_rays", r);
            rc = c5_synchronize(_ctx[r]);
        }
        check(rc, "trace_rays", r);
    }
    const std::size_t row_floats = _x * 2, tile_floats = row_floats * kTileRows;
    const std::size_t row_bytes = row_floats * sizeof(float), tile_bytes = tile_floats * sizeof(float);
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
        for (int gt = 0; gt < n_tiles && rc == ncclSuccess; ++gt) {
            const int r = gt % world, lt = gt / world;
            if (r == 0) continue;
            const std::size_t count = rows_of_tile(gt) * row_floats;
            rc = nccl.Send(m.strip[static_cast<std::size_t>(r)] + tile_floats * lt, count, ncclFloat, 0,
                          m.comm[static_cast<std::size_t>(r)], m.stream[static_cast<std::size_t>(r)]);
            if (rc == ncclSuccess)
                rc = nccl.Recv(m.root_frame + tile_floats * gt, count, ncclFloat, r, m.comm[0], m.stream[0]);
        }
        const ncclResult_t end = nccl.GroupEnd();
        if (rc != ncclSuccess || end != ncclSuccess)
            throw std::runtime_error(std::string("RCCL exchange failed: ") + nccl.GetErrorString(rc != ncclSuccess ? rc : end));
        hip_check(hipSetDevice(_devices[0]), "hipSetDevice");
        copy_own_tiles(0, m.stream[0]);
    } else {
        for (std::size_t r = 0; r < _ctx.size(); ++r) {
            hip_check(hipSetDevice(_devices[r]), "hipSetDevice");
            copy_own_tiles(r, m.stream[r]);  // runs on the sending device's stream, writes the root's memory
            if (r > 0) hip_check(hipEventRecord(m.done[r], m.stream[r]), "hipEventRecord");
        }
        hip_check(hipSetDevice(_devices[0]), "hipSetDevice");
        for (std::size_t r = 1; r < _ctx.size(); ++r) hip_check(hipStreamWaitEvent(m.stream[0], m.done[r], 0), "hipStreamWaitEvent");
    }
    hip_check(hipMemcpyAsync(f.image.get(), m.root_frame, _x * _y * 2 * sizeof(float), hipMemcpyDeviceToHost, m.stream[0]),
              "hipMemcpyAsync to the host");
    hip_check(hipStreamSynchronize(m.stream[0]), "hipStreamSynchronize");
}

object2d plane::trace_rays(tetra_value value_alpha, tetra_value value_Q) {
    if (value_alpha != tetra_value::alpha || value_Q != tetra_value::Q)
        throw std::runtime_error("trace_rays: only (alpha, Q) is supported");
    if (_flight.empty()) find_intersections();
    if (_exchange != exchange_mode::host) {
        finish_exchange(_flight.front());
    } else {
        for (int attempt = 0;; ++attempt) {
            bool retry = false;
            for (std::size_t r = 0; r < _ctx.size(); ++r) {
                const int rc = c5_render_host_wait(_ctx[r]);
                if (rc == C5_RETRY)
                    retry = true;
                else
                    check(rc, "trace_rays", r);
            }
            if (!retry) break;
            // An internal buffer was too small for this frame: it and every frame enqueued since are suspect
            // on every device.  Let them all finish, then render them again, in order, each with its own views.
            if (attempt >= 3) throw std::runtime_error("trace_rays: frames kept being reported incomplete");
            ++_retries;
            for (std::size_t k = 1; k < _flight.size(); ++k)
                for (std::size_t r = 0; r < _ctx.size(); ++r) {
                    const int rc = c5_render_host_wait(_ctx[r]);
                    if (rc != C5_RETRY) check(rc, "trace_rays", r);
                }
            for (frame_t& f : _flight) {
                send_views(f.views);
                start(f);
            }
            send_views(_views);
        }
    }
    frame_t f = std::move(_flight.front());
    _flight.pop_front();
    return object2d(std::move(f.image), _x, _y);
}

c5_stats plane::stats() {
    c5_stats sum{};
    for (std::size_t r = 0; r < _ctx.size(); ++r) {
        c5_stats st{};
        int rc = c5_get_stats(_ctx[r], &st);
        if (rc == C5_RETRY) rc = C5_OK;  // frames in flight behind the last completed one: theirs to report
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
