// `plane` and `object2d` with the reference's public interface (plane.hpp:17-33, object2d.hpp:13-22),
// implemented on the C ABI of libcourse5_hip.so instead of the OpenMP pixel loops.
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "course5_hip.h"
#include "scene.hpp"

// Two-channel fp32 image, row-major [y][x][2] (the order export_to_vti writes, object2d.cpp:17-21).
class object2d {
public:
    object2d() = default;
    object2d(std::vector<float> pixels, std::size_t res_x, std::size_t res_y)
        : _pixels(std::move(pixels)), _res_x(res_x), _res_y(res_y) {}
    void export_to_vti(const std::string& filename) const;  // object2d.cpp:7-29
    float at(std::size_t x, std::size_t y, std::size_t channel) const { return _pixels[(y * _res_x + x) * 2 + channel]; }
    std::size_t res_x() const { return _res_x; }
    std::size_t res_y() const { return _res_y; }
    const std::vector<float>& data() const { return _pixels; }

private:
    std::vector<float> _pixels;
    std::size_t _res_x = 0, _res_y = 0;
};

class plane {
public:
    plane() = delete;
    // plane.cpp:260-315.  objects3d: volume grids (transparent) and solids, in the reference's order;
    // global_boundaries = {x_max, x_min, y_max, y_min} (required here: the reference's automatic
    // bounding box is never used by its own main, main.cpp:83,127).
    explicit plane(std::size_t res_x, std::size_t res_y, std::vector<object3d_base> objects3d,
                   std::vector<double> global_boundaries = {}, int device = 0);
    ~plane();
    plane(const plane&) = delete;
    plane& operator=(const plane&) = delete;

    // plane.cpp:184-192: starts the frame on the GPU (view transform, records, entries, solids, walk)
    void find_intersections();
    // plane.cpp:144-172: waits for the frame and returns the image.  The two signatures select the
    // cell values for ch0/ch1 as in the reference; only (alpha, Q) is meaningful there and here.
    object2d trace_rays(tetra_value value_alpha, tetra_value value_Q);
    std::size_t count_all_intersections();  // plane.cpp:3-12 (segments of the last frame)

    std::size_t get_x() const { return _x; }
    std::size_t get_y() const { return _y; }

    // Re-send the objects' rotation lists (a sweep changes only these; the grid stays on the GPU).
    void update_views(std::vector<object3d_base>& objects3d);
    c5_stats stats();

private:
    void check(int rc, const char* what);
    c5_context* _ctx = nullptr;
    std::size_t _x = 0, _y = 0;
    std::vector<int> _slot_of_object;  // -1: part of the volume grid, >= 0: solid slot
    void* _device_image = nullptr;
    bool _in_flight = false;
};
