#include "plane.hpp"

#include <hip/hip_runtime_api.h>

#include <stdexcept>

#include "vtk_io.hpp"

void object2d::export_to_vti(const std::string& filename) const {
    write_vti(filename, _pixels.data(), static_cast<int>(_res_x), static_cast<int>(_res_y),
              !app::instance().config.raw_vti);
}

void plane::check(int rc, const char* what) {
    if (rc == C5_OK) return;
    // the reference throws std::runtime_error from the same places (plane.cpp:40,152,263,270)
    throw std::runtime_error(std::string(what) + ": " + c5_last_error(_ctx));
}

plane::plane(std::size_t res_x, std::size_t res_y, std::vector<object3d_base> objects3d,
             std::vector<double> global_boundaries, int device) {
    if (!global_boundaries.empty() && global_boundaries.size() != 4)
        throw std::runtime_error("plane initializer. wrong manual boundaries");  // plane.cpp:262-264
    if (objects3d.empty())
        throw std::runtime_error("plane initializer. empty set of objects to render");  // plane.cpp:269-271
    if (global_boundaries.empty())
        throw std::runtime_error("plane initializer. automatic boundaries are not supported: pass {x_max, x_min, y_max, y_min}");
    _x = res_x;
    _y = res_y;

    const int rc = c5_create(device, &_ctx);
    if (rc != C5_OK) throw std::runtime_error(std::string("c5_create: ") + c5_last_error(nullptr));

    // volume grids are merged into one indexed grid (the reference concatenates tetra vectors,
    // plane.cpp:290-293); solids keep one slot each, in order
    std::vector<double> pts, a, q;
    std::vector<int32_t> cells;
    int next_slot = 0;
    for (object3d_base& obj : objects3d) {
        const object3d_data& d = *obj.get_pointer();
        if (d.kind == tetra_type::solid) {
            if (next_slot >= C5_MAX_SOLIDS) throw std::runtime_error("too many solid objects");
            check(c5_set_solid(_ctx, next_slot, d.soup.data(), static_cast<int64_t>(d.soup.size() / 12), d.colour),
                  "c5_set_solid");
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
    if (!cells.empty())
        check(c5_upload_grid(_ctx, pts.data(), static_cast<int64_t>(pts.size() / 3), cells.data(),
                             static_cast<int64_t>(cells.size() / 4), a.data(), q.data()),
              "c5_upload_grid");
    check(c5_set_image(_ctx, static_cast<int>(res_x), static_cast<int>(res_y), global_boundaries.data()), "c5_set_image");
    check(c5_set_alpha_limit(_ctx, app::instance().config.limit_alpha_value), "c5_set_alpha_limit");  // line.cpp:204
    if (app::instance().config.reference_algorithm) check(c5_set_option(_ctx, "algorithm", 1.0), "c5_set_option");
    update_views(objects3d);
    if (hipMalloc(&_device_image, res_x * res_y * 2 * sizeof(float)) != hipSuccess)
        throw std::runtime_error("hipMalloc of the output image failed");
}

plane::~plane() {
    if (_ctx) {
        c5_synchronize(_ctx);
        c5_destroy(_ctx);
    }
    if (_device_image) (void)hipFree(_device_image);
}

void plane::update_views(std::vector<object3d_base>& objects3d) {
    bool grid_view_set = false;
    for (std::size_t k = 0; k < objects3d.size() && k < _slot_of_object.size(); ++k) {
        const object3d_data& d = *objects3d[k].get_pointer();
        const int n = static_cast<int>(d.rotations.size());
        if (_slot_of_object[k] < 0) {
            if (!grid_view_set) check(c5_set_view(_ctx, d.rotations.data(), n), "c5_set_view");
            grid_view_set = true;
        } else {
            check(c5_set_solid_view(_ctx, _slot_of_object[k], d.rotations.data(), n), "c5_set_solid_view");
        }
    }
}

void plane::find_intersections() {
    int rc = c5_render_device(_ctx, _device_image);
    check(rc, "find_intersections");
    _in_flight = true;
}

object2d plane::trace_rays(tetra_value value_alpha, tetra_value value_Q) {
    if (value_alpha != tetra_value::alpha || value_Q != tetra_value::Q)
        throw std::runtime_error("trace_rays: only (alpha, Q) is supported");
    if (!_in_flight) find_intersections();
    int rc = c5_synchronize(_ctx);
    for (int attempt = 0; rc == C5_RETRY && attempt < 3; ++attempt) {  // an internal buffer grew: redo the frame
        check(c5_render_device(_ctx, _device_image), "trace_rays");
        rc = c5_synchronize(_ctx);
    }
    check(rc, "trace_rays");
    _in_flight = false;
    std::vector<float> pixels(_x * _y * 2);
    if (hipMemcpy(pixels.data(), _device_image, pixels.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        throw std::runtime_error("copying the image from the GPU failed");
    return object2d(std::move(pixels), _x, _y);
}

c5_stats plane::stats() {
    c5_stats st{};
    check(c5_get_stats(_ctx, &st), "c5_get_stats");
    return st;
}

std::size_t plane::count_all_intersections() { return static_cast<std::size_t>(stats().segments); }
