// TEST INFRASTRUCTURE — inspection build for golden vector G3: the per-pixel segment
// lists of the REFERENCE's own line::calculate_intersections (line.cpp:84-148), i.e. for chosen pixels the
// sequence (tetra id, delta z) in the order std::sort leaves it, read from line::_intersections_delta.
// That member is private; this translation unit — and only this one — includes the reference's unmodified
// line.hpp with `private` defined as `public` around that one include, and is linked with the reference's
// line.o / tetra.o built the normal way.  Access specifiers do not change the layout g++ gives the class.
// Nothing of the reference is copied; only tests/ and tests/golden/make_golden.py call this.
#include <array>
#include <bitset>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <utility>
#include <vector>

#include <config.hpp>
#include <tetra.hpp>
#define private public  // inspection only: line::_intersections_delta
#include <line.hpp>
#undef private

#include "scan.hpp"

extern "C" {

// Same scene arguments as c5r_render.  probe_ij[n_probe][2] = (col, row); out[n_probe][cap][2] = (tetra id,
// delta z) as doubles; count[n_probe] = segments of the pixel.
int c5r_probe(const double* xyz, int64_t n_pts, const int32_t* cell_vert, int64_t n_cells, const double* alpha,
              const double* q, const double* rots, int n_rot, int res_x, int res_y, const double* bounds4,
              const int32_t* probe_ij, int n_probe, int cap, double* out, int32_t* count, char* err, int errlen) {
    try {
        std::vector<tetra> tets;
        tets.reserve(static_cast<size_t>(n_cells));
        for (int64_t c = 0; c < n_cells; ++c) {
            std::array<std::array<double, 3>, 4> pts{};
            for (int v = 0; v < 4; ++v) {
                const int32_t pid = cell_vert[4 * c + v];
                if (pid < 0 || pid >= n_pts) throw std::runtime_error("cell references a point id out of range");
                for (int k = 0; k < 3; ++k) pts[static_cast<size_t>(v)][static_cast<size_t>(k)] = xyz[3 * pid + k];
            }
            tetra t(pts, alpha[c], q[c], tetra_type::transparent);
            for (int r = 0; r < n_rot; ++r) {  // main.cpp:105-107 through the reference's own rotations
                if (rots[3 * r] == 0.0)
                    t.rotate_around_x_axis(rots[3 * r + 1]);
                else
                    t.rotate_around_y_axis(rots[3 * r + 1], rots[3 * r + 2]);
            }
            tets.push_back(t);
        }
        c5scan::PixelGrid grid;
        grid.init(static_cast<size_t>(res_x), static_cast<size_t>(res_y), bounds4);
        std::map<std::pair<size_t, size_t>, std::unique_ptr<line>> lines;
        for (int k = 0; k < n_probe; ++k) {
            const size_t i = static_cast<size_t>(probe_ij[2 * k]), j = static_cast<size_t>(probe_ij[2 * k + 1]);
            if (i >= grid.res_x || j >= grid.res_y) throw std::runtime_error("probe outside the image");
            lines[{i, j}] = std::make_unique<line>(grid.X[i], grid.Y[j]);
        }
        for (size_t id = 0; id < tets.size(); ++id) {
            const tetra& t = tets[id];
            for (size_t f = 0; f < 4; ++f) {
                const int* fv = c5scan::kFaceVerts[f];
                c5scan::scan_face(grid, t[static_cast<size_t>(fv[0])].data(), t[static_cast<size_t>(fv[1])].data(),
                                  t[static_cast<size_t>(fv[2])].data(), [&](size_t i, size_t j) {
                                      auto it = lines.find({i, j});
                                      if (it != lines.end()) it->second->add_tetra_intersection(id, f, 0);
                                  });
            }
        }
        for (int k = 0; k < n_probe; ++k) {
            line& L = *lines[{static_cast<size_t>(probe_ij[2 * k]), static_cast<size_t>(probe_ij[2 * k + 1])}];
            L.calculate_intersections(tets);                       // line.cpp:84-148
            const std::vector<intersection_data>& seg = L._intersections_delta;  // private in the reference
            count[k] = static_cast<int32_t>(seg.size());
            for (size_t s = 0; s < seg.size() && s < static_cast<size_t>(cap); ++s) {
                out[2 * (static_cast<size_t>(k) * cap + s)] = static_cast<double>(seg[s].tetra_id);
                out[2 * (static_cast<size_t>(k) * cap + s) + 1] = seg[s].delta_z;
            }
        }
        return 0;
    } catch (const std::exception& e) {
        if (err && errlen > 0) std::snprintf(err, static_cast<size_t>(errlen), "%s", e.what());
        return 1;
    }
}

}  // extern "C"
