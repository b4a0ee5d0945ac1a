// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// CPU restatement of the reference's pixel grid and face scan conversion
// (reference: project/src/plane.cpp).  Shared by oracle.cpp (full restatement)
// and ref_driver.cpp (which feeds the *real* reference `line`/`tetra` classes).
// plane.cpp itself cannot be compiled in this image: plane.hpp pulls in
// object3d_base.hpp / object2d.hpp which need VTK headers (absent), so this part
// of the path is restated, not linked.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// anything under oracle/.
#pragma once

#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace c5scan {

// Pixel grid of the image plane.  Reference: plane::plane, plane.cpp:260-315.
struct PixelGrid {
    size_t res_x = 0, res_y = 0;
    // bounds in the reference's order {x_max, x_min, y_max, y_min} (main.cpp:83)
    double b[4] = {0, 0, 0, 0};
    double step_x = 0, step_y = 0;
    std::vector<double> X, Y;  // pixel ray coordinates

    void init(size_t rx, size_t ry, const double bounds[4]) {
        res_x = rx;
        res_y = ry;
        for (int k = 0; k < 4; ++k) b[k] = bounds[k];
        // plane.cpp:298-302
        const double span_x = b[0] - b[1];
        const double span_y = b[2] - b[3];
        step_x = span_x / (static_cast<double>(rx) - 1.);
        step_y = span_y / (static_cast<double>(ry) - 1.);
        // plane.cpp:304-314 — coordinates are accumulated, not i*step.
        X.resize(rx);
        Y.resize(ry);
        double cx = b[1];
        for (size_t i = 0; i < rx; ++i) {
            X[i] = cx;
            cx = cx + step_x;
        }
        double cy = b[3];
        for (size_t j = 0; j < ry; ++j) {
            Y[j] = cy;
            cy = cy + step_y;
        }
    }

    // plane::get_pixel_by_x, plane.cpp:194-202 (clamped fractional index)
    double frac_x(double x) const {
        double r = (x - b[1]) / step_x;
        const double hi = static_cast<double>(res_x) - 1;
        if (r < 0) return 0;
        if (r > hi) return hi;
        return r;
    }
    // plane::get_pixel_by_y, plane.cpp:204-212
    double frac_y(double y) const {
        double r = (y - b[3]) / step_y;
        const double hi = static_cast<double>(res_y) - 1;
        if (r < 0) return 0;
        if (r > hi) return hi;
        return r;
    }
};

// plane::line_common_eq, plane.cpp:46-48: signed edge function of `pos` w.r.t. a->b.
inline double edge_side(const double* a, const double* b, const double* pos) {
    return (b[1] - a[1]) * pos[0] + (a[0] - b[0]) * pos[1] + (b[0] * a[1] - a[0] * b[1]);
}

// plane::line_rev_function_eq, plane.cpp:50-55: x on the edge a-b at height y.
inline double edge_x_at(const double* a, const double* b, double y) {
    if (std::fabs(a[1] - b[1]) < DBL_EPSILON) return a[0];
    return (a[0] - b[0]) * (y - a[1]) / (a[1] - b[1]) + a[0];
}

// Inclusive scanline coverage of one projected triangle.
// Reference: plane::find_intersections_with_polygon, plane.cpp:57-142.
// `emit(i, j)` is called once per covered pixel, rows ascending, columns ascending.
// Returns the number of covered pixels.
template <class Emit>
inline size_t scan_face(const PixelGrid& g, const double* v0, const double* v1, const double* v2,
                        Emit&& emit) {
    std::array<const double*, 3> p{v0, v1, v2};
    // plane.cpp:61 — descending y
    std::sort(p.begin(), p.end(), [](const double* a, const double* b) { return a[1] > b[1]; });

    // plane.cpp:66-89 — which side of the long edge p0-p2 the middle vertex lies on
    const double side = edge_side(p[0], p[2], p[1]);
    const bool up_left = (p[0][0] >= p[2][0]) && (side >= 0);
    const bool down_right = (p[0][0] < p[2][0]) && (side > 0);
    const bool long_edge_is_left = !(up_left || down_right);

    // plane.cpp:91-100 — row range; size_t conversions as in the reference
    const size_t row_hi = static_cast<size_t>(std::floor(g.frac_y(p[0][1])));
    const size_t row_lo = static_cast<size_t>(std::ceil(g.frac_y(p[2][1])));

    size_t n = 0;
    size_t row = row_lo;
    double y = g.Y[row];  // plane.cpp:100 (_lines[0][row].y())
    for (; row <= row_hi; ++row) {
        // plane.cpp:106-122
        const double* lower_a = (y < p[1][1]) ? p[2] : p[0];
        const double x_long = edge_x_at(p[0], p[2], y);
        const double x_short = edge_x_at(lower_a, p[1], y);
        const double x_lo = long_edge_is_left ? x_long : x_short;
        const double x_hi = long_edge_is_left ? x_short : x_long;

        // plane.cpp:126-127
        const size_t col_hi = static_cast<size_t>(std::floor(g.frac_x(x_hi)));
        const size_t col_lo = static_cast<size_t>(std::ceil(g.frac_x(x_lo)));
        for (size_t col = col_lo; col <= col_hi; ++col) {
            emit(col, row);
            ++n;
        }
        y = y + g.step_y;  // plane.cpp:138
    }
    return n;
}

// Face numbering of a tetrahedron: plane.cpp:16-21, 30-37.
static const int kFaceVerts[4][3] = {{0, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 2, 3}};

}  // namespace c5scan
