// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// Driver around the REAL reference translation units project/src/line.cpp and
// project/src/tetra.cpp, which oracle/Makefile compiles in place from /root/reference
// into oracle/_ref/ (never copied into this repository).  Everything arithmetic on the
// per-pixel path — vertex rotation (tetra::rotate_around_*), hit pairing
// (line::add_tetra_intersection), face z (line::find_polygon_intersection_z via
// calculate_intersections), the std::sort, tau (direct_calculate_ray_value) and I
// (integrate_ray_value_by_i) — is executed by the reference's own object code.
//
// plane.cpp / main.cpp / object3d_base.cpp need VTK or Boost headers that this image does
// not have, so the pixel grid and face scan conversion come from the restatement in
// scan.hpp, and the loops of plane::find_intersections / plane::trace_rays
// (plane.cpp:144-192) are re-driven here.  This driver runs serially (-j1, thread id 0).
//
// Used to (a) validate oracle.cpp and (b) generate tests/golden/*.npz.
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

#include <line.hpp>   // reference header (-I/root/reference/project/include)
#include <tetra.hpp>  // reference header

#include "scan.hpp"

namespace {

void set_err(char* err, int errlen, const char* msg) {
    if (err && errlen > 0) std::snprintf(err, static_cast<size_t>(errlen), "%s", msg);
}

tetra make_tet(const double* p12, double v1, double v2, tetra_type kind) {
    std::array<std::array<double, 3>, 4> pts{};
    for (int v = 0; v < 4; ++v)
        for (int k = 0; k < 3; ++k) pts[static_cast<size_t>(v)][static_cast<size_t>(k)] = p12[3 * v + k];
    return tetra(pts, v1, v2, kind);
}

void rotate_tet(tetra& t, const double* rots, int n_rot) {
    for (int r = 0; r < n_rot; ++r) {
        const double* R = rots + 3 * r;
        if (R[0] == 0.0)
            t.rotate_around_x_axis(R[1]);
        else
            t.rotate_around_y_axis(R[1], R[2]);
    }
}

}  // namespace

extern "C" {

// Rotate n points through the reference's tetra::rotate_around_* (4 points per tetra object).
void c5r_rotate_points(double* xyz, int64_t n, const double* rots, int n_rot) {
    for (int64_t base = 0; base < n; base += 4) {
        double p12[12] = {0};
        const int64_t m = std::min<int64_t>(4, n - base);
        std::memcpy(p12, xyz + 3 * base, sizeof(double) * 3 * static_cast<size_t>(m));
        tetra t = make_tet(p12, 0, 0, tetra_type::transparent);
        rotate_tet(t, rots, n_rot);
        for (int64_t v = 0; v < m; ++v)
            for (int k = 0; k < 3; ++k)
                xyz[3 * (base + v) + k] = t[static_cast<size_t>(v)][static_cast<size_t>(k)];
    }
}

// Same contract as c5o_render (oracle.cpp) minus timing/probes.
int c5r_render(const double* xyz, int64_t n_pts, const int32_t* cell_vert, int64_t n_cells,
               const double* alpha, const double* q, const double* rots, int n_rot,
               const double* solid_tets, const double* solid_colour, int64_t n_solid, int res_x,
               int res_y, const double* bounds4, double alpha_limit, float* out, int64_t* stats,
               char* err, int errlen) {
    try {
        if (res_x < 2 || res_y < 2) throw std::runtime_error("critical error. empty plane");
        app::instance().config.limit_alpha_value = alpha_limit;  // read at line.cpp:204

        std::vector<tetra> tets;
        tets.reserve(static_cast<size_t>(n_cells + n_solid));
        for (int64_t c = 0; c < n_cells; ++c) {
            double p12[12];
            for (int v = 0; v < 4; ++v) {
                const int32_t pid = cell_vert[4 * c + v];
                if (pid < 0 || pid >= n_pts)
                    throw std::runtime_error("cell references a point id out of range");
                std::memcpy(p12 + 3 * v, xyz + 3 * pid, sizeof(double) * 3);
            }
            tetra t = make_tet(p12, alpha[c], q[c], tetra_type::transparent);
            rotate_tet(t, rots, n_rot);
            tets.push_back(t);
        }
        for (int64_t s = 0; s < n_solid; ++s)
            tets.push_back(make_tet(solid_tets + 12 * s, solid_colour[s], 0, tetra_type::solid));

        c5scan::PixelGrid grid;
        grid.init(static_cast<size_t>(res_x), static_cast<size_t>(res_y), bounds4);

        // reference `line` objects, indexed [x][y] like plane.hpp:59-62
        std::vector<std::vector<line>> lines(grid.res_x);
        for (size_t i = 0; i < grid.res_x; ++i) {
            lines[i].reserve(grid.res_y);
            for (size_t j = 0; j < grid.res_y; ++j) lines[i].push_back(line(grid.X[i], grid.Y[j]));
        }

        // plane::find_intersections loop (plane.cpp:184-192, 14-44), thread id 0
        for (size_t id = 0; id < tets.size(); ++id) {
            const tetra& t = tets[id];
            const bool solid = t.get_tetra_type() == tetra_type::solid;
            const double colour = solid ? t.access_value(tetra_value::solid_color) : 0.0;
            size_t hits = 0;
            for (size_t f = 0; f < 4; ++f) {
                const int* fv = c5scan::kFaceVerts[f];
                hits += c5scan::scan_face(
                    grid, t[static_cast<size_t>(fv[0])].data(), t[static_cast<size_t>(fv[1])].data(),
                    t[static_cast<size_t>(fv[2])].data(), [&](size_t i, size_t j) {
                        if (solid)
                            lines[i][j].mark_solid_color(colour);
                        else
                            lines[i][j].add_tetra_intersection(id, f, 0);
                    });
            }
            if (hits % 2 == 1)
                throw std::runtime_error("critical error. odd number of intersections");
        }

        // plane::trace_rays loop (plane.cpp:161-169)
        int64_t S = 0, covered = 0;
        for (size_t i = 0; i < grid.res_x; ++i) {
            for (size_t j = 0; j < grid.res_y; ++j) {
                line& L = lines[i][j];
                const size_t n = L.number_of_intersections();
                S += static_cast<int64_t>(n);
                if (n) ++covered;
                L.calculate_intersections(tets);
                float* o = out + 2 * (j * grid.res_x + i);
                o[0] = static_cast<float>(L.direct_calculate_ray_value(tets, tetra_value::alpha));
                o[1] = static_cast<float>(
                    L.integrate_ray_value_by_i(tets, tetra_value::alpha, tetra_value::Q));
                L.free_memory();
            }
        }
        if (stats) {
            stats[0] = S;
            stats[1] = covered;
            stats[2] = -1;
        }
        return 0;
    } catch (const std::exception& e) {
        set_err(err, errlen, e.what());
        return 1;
    }
}

const char* c5r_kind() { return "reference"; }

}  // extern "C"
