// TEST INFRASTRUCTURE — NOT PRODUCT CODE.  (CPU oracle for the render hot path.)
//
// Own-words CPU restatement of the reference render path of mlozhechko/course5:
//   view transform          tetra.cpp:44-62, object3d_base.cpp:202-219, main.cpp:96-107
//   pixel grid + binning    plane.cpp:14-44, 57-142, 184-212, 260-315   (scan.hpp)
//   per-pixel hit pairing   line.cpp:29-67, 229-232, line.hpp:71-85
//   per-pixel resolve       line.cpp:84-148 (z of faces, sort), 150-174 (plane eq.)
//   ch0 "tau"               line.cpp:176-193
//   ch1 "I"                 line.cpp:195-227
//   fp32 narrowing          plane.cpp:145-146, 165-166
//
// Pinning: the line.cpp / tetra.cpp arithmetic restated here is checked bit-for-bit
// against the real reference translation units (oracle/_ref, built by oracle/Makefile
// from /root/reference in place) by tests/test_oracle_vs_ref.py and through the committed
// golden vectors in tests/golden/.  plane.cpp cannot be built in this image (needs VTK
// headers); its restatement (scan.hpp) is shared by both builds and is pinned only by the
// segment counts the survey recorded from the full reference.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
#include <omp.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>

#include "scan.hpp"

namespace {

using c5scan::kFaceVerts;
using c5scan::PixelGrid;

// tetra.hpp:42-45 — AoS record, private vertex copies per cell
struct Tet {
    double p[4][3];
    double val[2];  // [0] = alpha or solid colour, [1] = Q  (tetra.hpp:8)
    int solid;      // tetra.hpp:10
};

// tetra.cpp:44-48
inline void spin_about_x(double a, double* pt) {
    const double y0 = pt[1];
    pt[1] = pt[1] * std::cos(a) - pt[2] * std::sin(a);
    pt[2] = y0 * std::sin(a) + pt[2] * std::cos(a);
}
// tetra.cpp:51-62
inline void spin_about_y(double a, double* pt, double x0) {
    pt[0] -= x0;
    const double xs = pt[0];
    pt[0] = pt[0] * std::cos(a) - pt[2] * std::sin(a);
    pt[2] = xs * std::sin(a) + pt[2] * std::cos(a);
    pt[0] += x0;
}

// rots: [n][3] rows of {axis (0 = x, 1 = y), angle, x0}
inline void apply_rotations(double* pt, const double* rots, int n_rot) {
    for (int r = 0; r < n_rot; ++r) {
        const double* R = rots + 3 * r;
        if (R[0] == 0.0)
            spin_about_x(R[1], pt);
        else
            spin_about_y(R[1], pt, R[2]);
    }
}

constexpr uint32_t kIdMask = 0x0FFFFFFFu;  // line.cpp:27
constexpr int kFaceShift = 28;             // line.cpp:19

// line.cpp:150-174 — z where the vertical line (x, y) pierces the plane through a, b, c
inline double z_on_face(double x, double y, const double* a, const double* b, const double* c) {
    const double tx = (x - a[0]) * ((b[1] - a[1]) * (c[2] - a[2]) - (c[1] - a[1]) * (b[2] - a[2]));
    const double ty = (y - a[1]) * ((b[0] - a[0]) * (c[2] - a[2]) - (c[0] - a[0]) * (b[2] - a[2]));
    const double den = ((b[0] - a[0]) * (c[1] - a[1]) - (c[0] - a[0]) * (b[1] - a[1]));
    return (ty - tx) / den + a[2];
}

struct Segment {
    double z_hi;
    double dz;
    uint32_t tet;
};

// line.cpp:84-148
inline void resolve_pixel(double x, double y, const std::vector<uint32_t>& words,
                          const std::vector<Tet>& tets, std::vector<Segment>& out) {
    out.clear();
    out.reserve(words.size());
    for (uint32_t w : words) {
        const uint32_t id = w & kIdMask;
        const Tet& t = tets.at(id);  // line.cpp:102 (bounds-checked)
        double z[2] = {0, 0};
        int k = 0;
        // bits 31..28 <-> faces 3..0, evaluated in that order (line.cpp:103-122)
        for (int f = 3; f >= 0; --f) {
            if (w & (1u << (kFaceShift + f))) {
                if (k >= 2) throw std::runtime_error("oracle: more than two faces flagged");
                const int* fv = kFaceVerts[f];
                z[k++] = z_on_face(x, y, t.p[fv[0]], t.p[fv[1]], t.p[fv[2]]);
            }
        }
        if (z[0] < z[1]) std::swap(z[0], z[1]);  // line.cpp:124-128
        out.push_back(Segment{z[0], z[0] - z[1], id});
    }
    // line.cpp:138 (std::sort, descending key, unstable)
    std::sort(out.begin(), out.end(),
              [](const Segment& a, const Segment& b) { return a.z_hi > b.z_hi; });
}

// line.cpp:176-193
inline double tau_of(const std::vector<Segment>& segs, const std::vector<Tet>& tets) {
    double sum = 0;
    for (const Segment& s : segs) sum = sum + s.dz * tets[s.tet].val[0];
    return sum;
}

// one step of line.cpp:206-225
inline double emission_step(double I, double alpha, double Q, double dz, double alpha_limit) {
    if (alpha > alpha_limit) alpha = alpha_limit;
    const double C = Q - alpha * I;
    if (alpha < std::numeric_limits<double>::epsilon()) return I;
    return (Q - C * std::exp(-alpha * dz)) / alpha;
}

// line.cpp:195-227
inline double intensity_of(const std::vector<Segment>& segs, const std::vector<Tet>& tets,
                           double alpha_limit) {
    double I = 0;
    for (ptrdiff_t k = static_cast<ptrdiff_t>(segs.size()) - 1; k >= 0; --k) {
        const Tet& t = tets[segs[k].tet];
        I = emission_step(I, t.val[0], t.val[1], segs[k].dz, alpha_limit);
    }
    return I;
}

struct PixelState {
    std::vector<uint32_t> words;  // line.hpp:81
    bool marked = false;          // line.hpp:89
    double mark = 0;              // line.hpp:90
};

struct Scene {
    PixelGrid grid;
    std::vector<Tet> tets;
    std::vector<PixelState> px;  // [x][y] like plane.hpp:59-62 -> index x*res_y + y
    int threads = 1;
    // Test-side economy, not reference behaviour: keep only rows j with j % row_stride == row_phase (their
    // pixels are binned, resolved and written exactly as in a full render — pixels are independent,
    // plane.cpp:161-169; the other rows stay zero).  Lets a 4800x3600 frame be checked on every k-th row.
    int row_stride = 1, row_phase = 0;
    bool keeps(size_t j) const { return row_stride <= 1 || static_cast<int>(j % static_cast<size_t>(row_stride)) == row_phase; }
    // line.hpp:84-85: per-pixel, per-thread pairing word; flag == (word != 0)
    std::vector<std::vector<uint32_t>> pend;  // [thread][pixel]
    std::vector<std::mutex> locks;            // striped stand-in for line.hpp:87
    static constexpr size_t kLockStripes = 1 << 14;

    size_t pix(size_t i, size_t j) const { return i * grid.res_y + j; }

    // line.cpp:29-67
    void add_hit(size_t i, size_t j, uint32_t id, int face, int tid) {
        if (!keeps(j)) return;
        PixelState& P = px[pix(i, j)];
        if (P.marked) return;
        uint32_t& buf = pend[tid][pix(i, j)];
        const bool second = (buf != 0);
        buf |= (1u << (kFaceShift + face));
        buf |= id;
        if (!second) {
            if ((buf & kIdMask) != id)  // unreachable with flag == (word != 0); kept for parity
                throw std::runtime_error("tetrahedron intersection fatal data error");
        } else {
            if ((buf & kIdMask) != id) {
                // the reference would push a corrupted id here (line.cpp:49-52) and fail later
                // in tetra_vector.at(); surface it directly
                throw std::runtime_error("tetrahedron intersection fatal data error");
            }
            std::lock_guard<std::mutex> lk(locks[pix(i, j) & (kLockStripes - 1)]);
            P.words.push_back(buf);  // line.cpp:229-232
            buf = 0;
        }
    }

    // plane.cpp:14-44
    size_t bin_tet(uint32_t id, int tid) {
        const Tet& t = tets[id];
        size_t hits = 0;
        for (int f = 0; f < 4; ++f) {
            const int* fv = kFaceVerts[f];
            if (t.solid) {
                const double colour = t.val[0];
                hits += c5scan::scan_face(grid, t.p[fv[0]], t.p[fv[1]], t.p[fv[2]],
                                          [&](size_t i, size_t j) {
                                              if (!keeps(j)) return;
                                              PixelState& P = px[pix(i, j)];  // line.cpp:246-249
                                              P.marked = true;
                                              P.mark = colour;
                                          });
            } else {
                hits += c5scan::scan_face(grid, t.p[fv[0]], t.p[fv[1]], t.p[fv[2]],
                                          [&](size_t i, size_t j) { add_hit(i, j, id, f, tid); });
            }
        }
        if (hits % 2 == 1)  // plane.cpp:39-41
            throw std::runtime_error("critical error. odd number of intersections");
        return hits / 2;
    }
};

double now_ms() {
    using clk = std::chrono::steady_clock;
    return std::chrono::duration<double, std::milli>(clk::now().time_since_epoch()).count();
}

void set_err(char* err, int errlen, const char* msg) {
    if (err && errlen > 0) {
        std::snprintf(err, static_cast<size_t>(errlen), "%s", msg);
    }
}

}  // namespace

extern "C" {

// Apply the reference's sequential in-place rotations to n points (xyz rows).
// rots: [n_rot][3] = {axis, angle, x0}.  (tetra.cpp:44-62)
void c5o_rotate_points(double* xyz, int64_t n, const double* rots, int n_rot) {
    for (int64_t i = 0; i < n; ++i) apply_rotations(xyz + 3 * i, rots, n_rot);
}

// line.cpp:150-174 exposed for known-answer tests.
double c5o_face_z(double x, double y, const double* a, const double* b, const double* c) {
    return z_on_face(x, y, a, b, c);
}

// line.cpp:206-225 exposed for known-answer tests.
double c5o_emission_step(double I, double alpha, double Q, double dz, double alpha_limit) {
    return emission_step(I, alpha, Q, dz, alpha_limit);
}

// Pixel coordinate tables (plane.cpp:298-314).
void c5o_pixel_coords(int res_x, int res_y, const double* bounds4, double* X, double* Y) {
    PixelGrid g;
    g.init(static_cast<size_t>(res_x), static_cast<size_t>(res_y), bounds4);
    std::memcpy(X, g.X.data(), sizeof(double) * g.X.size());
    std::memcpy(Y, g.Y.data(), sizeof(double) * g.Y.size());
}

// One face through the scan conversion (plane.cpp:57-142 as restated in scan.hpp): the covered pixels
// (col, row) in emission order, for unit tests with hand-computed triangles.  Returns the count (which may
// exceed cap; only the first cap pairs are stored).
int64_t c5o_scan_face(int res_x, int res_y, const double* bounds4, const double* v0, const double* v1, const double* v2,
                      int32_t* out_ij, int64_t cap) {
    PixelGrid g;
    g.init(static_cast<size_t>(res_x), static_cast<size_t>(res_y), bounds4);
    int64_t n = 0;
    c5scan::scan_face(g, v0, v1, v2, [&](size_t i, size_t j) {
        if (n < cap) {
            out_ij[2 * n] = static_cast<int32_t>(i);
            out_ij[2 * n + 1] = static_cast<int32_t>(j);
        }
        ++n;
    });
    return n;
}

// Full render.
//   xyz[n_pts][3], cell_vert[n_cells][4], alpha[n_cells], q[n_cells]: the volume grid (raw);
//   rots[n_rot][3]: view rotations applied to every grid vertex copy (main.cpp:105-107);
//   solid_tets[n_solid][4][3] + solid_colour[n_solid]: already transformed solid tets, appended
//     after the grid tets (main.cpp:127, plane.cpp:290-293);
//   bounds4 = {x_max, x_min, y_max, y_min};
//   out[res_y][res_x][2] fp32 (object2d.cpp:17-21 order);
//   stats[0] = S (count_all_intersections, plane.cpp:3-12), [1] = pixels with >= 1 segment,
//     [2] = solid-marked pixels; timing_ms[0..2] = grid ctor / binning / resolve;
//   probe_ij/probe_out: optional per-pixel segment dump (n_probe pixels, up to probe_cap
//     segments each as {tet, z_hi, dz}); probe_count[n_probe].
int c5o_render_rows(const double* xyz, int64_t n_pts, const int32_t* cell_vert, int64_t n_cells,
                    const double* alpha, const double* q, const double* rots, int n_rot,
                    const double* solid_tets, const double* solid_colour, int64_t n_solid, int res_x,
                    int res_y, const double* bounds4, double alpha_limit, int threads, float* out,
                    int64_t* stats, double* timing_ms, const int32_t* probe_ij, int n_probe,
                    int probe_cap, double* probe_out, int32_t* probe_count, char* err, int errlen,
                    int row_stride, int row_phase);

int c5o_render(const double* xyz, int64_t n_pts, const int32_t* cell_vert, int64_t n_cells,
               const double* alpha, const double* q, const double* rots, int n_rot,
               const double* solid_tets, const double* solid_colour, int64_t n_solid, int res_x,
               int res_y, const double* bounds4, double alpha_limit, int threads, float* out,
               int64_t* stats, double* timing_ms, const int32_t* probe_ij, int n_probe,
               int probe_cap, double* probe_out, int32_t* probe_count, char* err, int errlen) {
    return c5o_render_rows(xyz, n_pts, cell_vert, n_cells, alpha, q, rots, n_rot, solid_tets, solid_colour, n_solid,
                           res_x, res_y, bounds4, alpha_limit, threads, out, stats, timing_ms, probe_ij, n_probe,
                           probe_cap, probe_out, probe_count, err, errlen, 1, 0);
}

// The same render restricted to the rows j with j % row_stride == row_phase (see Scene::keeps); the
// statistics then count those rows only.
int c5o_render_rows(const double* xyz, int64_t n_pts, const int32_t* cell_vert, int64_t n_cells,
                    const double* alpha, const double* q, const double* rots, int n_rot,
                    const double* solid_tets, const double* solid_colour, int64_t n_solid, int res_x,
                    int res_y, const double* bounds4, double alpha_limit, int threads, float* out,
                    int64_t* stats, double* timing_ms, const int32_t* probe_ij, int n_probe,
                    int probe_cap, double* probe_out, int32_t* probe_count, char* err, int errlen,
                    int row_stride, int row_phase) {
    try {
        if (res_x < 2 || res_y < 2) throw std::runtime_error("critical error. empty plane");
        if (n_cells + n_solid <= 0)
            throw std::runtime_error("plane initializer. empty set of objects to render");
        if (n_cells + n_solid > static_cast<int64_t>(kIdMask))
            throw std::runtime_error("tetrahedron id does not fit 28 bits");
        if (threads < 1) threads = 1;
        Scene sc;
        sc.threads = threads;
        sc.row_stride = row_stride < 1 ? 1 : row_stride;
        sc.row_phase = row_phase;

        // object3d_base.cpp:13-53 — per-cell vertex copies; main.cpp:105-107 — view transform
        sc.tets.resize(static_cast<size_t>(n_cells + n_solid));
#pragma omp parallel for num_threads(threads) schedule(static)
        for (int64_t c = 0; c < n_cells; ++c) {
            Tet& t = sc.tets[static_cast<size_t>(c)];
            for (int v = 0; v < 4; ++v) {
                const int32_t pid = cell_vert[4 * c + v];
                if (pid < 0 || pid >= n_pts) continue;  // reported below
                t.p[v][0] = xyz[3 * pid + 0];
                t.p[v][1] = xyz[3 * pid + 1];
                t.p[v][2] = xyz[3 * pid + 2];
                apply_rotations(t.p[v], rots, n_rot);
            }
            t.val[0] = alpha[c];
            t.val[1] = q[c];
            t.solid = 0;
        }
        for (int64_t c = 0; c < 4 * n_cells; ++c)
            if (cell_vert[c] < 0 || cell_vert[c] >= n_pts)
                throw std::runtime_error("cell references a point id out of range");
        for (int64_t s = 0; s < n_solid; ++s) {
            Tet& t = sc.tets[static_cast<size_t>(n_cells + s)];
            std::memcpy(t.p, solid_tets + 12 * s, sizeof(double) * 12);
            t.val[0] = solid_colour[s];
            t.val[1] = 0;
            t.solid = 1;
        }

        double t0 = now_ms();
        sc.grid.init(static_cast<size_t>(res_x), static_cast<size_t>(res_y), bounds4);
        const size_t n_px = sc.grid.res_x * sc.grid.res_y;
        sc.px.resize(n_px);
        sc.pend.assign(static_cast<size_t>(threads), std::vector<uint32_t>(n_px, 0u));
        sc.locks = std::vector<std::mutex>(Scene::kLockStripes);
        double t1 = now_ms();

        // plane.cpp:184-192
        std::string failure;
        const int64_t n_all = n_cells + n_solid;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 8)
        for (int64_t id = 0; id < n_all; ++id) {
            try {
                sc.bin_tet(static_cast<uint32_t>(id), omp_get_thread_num());
            } catch (const std::exception& e) {
#pragma omp critical
                failure = e.what();
            }
        }
        if (!failure.empty()) throw std::runtime_error(failure);
        double t2 = now_ms();

        // plane.cpp:144-172
        int64_t S = 0, covered = 0, marked = 0;
#pragma omp parallel num_threads(threads) reduction(+ : S, covered, marked)
        {
            std::vector<Segment> segs;
#pragma omp for schedule(dynamic, 8) collapse(2)
            for (size_t i = 0; i < sc.grid.res_x; ++i) {
                for (size_t j = 0; j < sc.grid.res_y; ++j) {
                    if (!sc.keeps(j)) continue;
                    PixelState& P = sc.px[sc.pix(i, j)];
                    float* o = out + 2 * (j * sc.grid.res_x + i);
                    try {
                        resolve_pixel(sc.grid.X[i], sc.grid.Y[j], P.words, sc.tets, segs);
                    } catch (const std::exception& e) {
#pragma omp critical
                        failure = e.what();
                        continue;
                    }
                    S += static_cast<int64_t>(P.words.size());
                    if (!P.words.empty()) ++covered;
                    if (P.marked) {
                        ++marked;
                        o[0] = static_cast<float>(P.mark);  // line.cpp:177-179
                        o[1] = static_cast<float>(P.mark);  // line.cpp:197-199
                    } else {
                        o[0] = static_cast<float>(tau_of(segs, sc.tets));
                        o[1] = static_cast<float>(intensity_of(segs, sc.tets, alpha_limit));
                    }
                    std::vector<uint32_t>().swap(P.words);  // line.cpp:241-244
                }
            }
        }
        if (!failure.empty()) throw std::runtime_error(failure);
        double t3 = now_ms();

        // inspection pass for selected pixels (re-bins serially; small fixtures only)
        if (n_probe > 0 && probe_ij && probe_out && probe_count) {
            std::vector<std::vector<uint32_t>> words(static_cast<size_t>(n_probe));
            for (int64_t id = 0; id < n_cells; ++id) {
                const Tet& t = sc.tets[static_cast<size_t>(id)];
                std::vector<uint32_t> pend(static_cast<size_t>(n_probe), 0u);
                for (int f = 0; f < 4; ++f) {
                    const int* fv = kFaceVerts[f];
                    c5scan::scan_face(sc.grid, t.p[fv[0]], t.p[fv[1]], t.p[fv[2]],
                                      [&](size_t i, size_t j) {
                                          for (int k = 0; k < n_probe; ++k) {
                                              if (static_cast<size_t>(probe_ij[2 * k]) != i ||
                                                  static_cast<size_t>(probe_ij[2 * k + 1]) != j)
                                                  continue;
                                              uint32_t& b = pend[static_cast<size_t>(k)];
                                              const bool second = b != 0;
                                              b |= (1u << (kFaceShift + f)) | static_cast<uint32_t>(id);
                                              if (second) {
                                                  words[static_cast<size_t>(k)].push_back(b);
                                                  b = 0;
                                              }
                                          }
                                      });
                }
            }
            std::vector<Segment> segs;
            for (int k = 0; k < n_probe; ++k) {
                resolve_pixel(sc.grid.X[static_cast<size_t>(probe_ij[2 * k])],
                              sc.grid.Y[static_cast<size_t>(probe_ij[2 * k + 1])],
                              words[static_cast<size_t>(k)], sc.tets, segs);
                probe_count[k] = static_cast<int32_t>(segs.size());
                for (int s = 0; s < static_cast<int>(segs.size()) && s < probe_cap; ++s) {
                    double* r = probe_out + 3 * (static_cast<size_t>(k) * probe_cap + s);
                    r[0] = static_cast<double>(segs[static_cast<size_t>(s)].tet);
                    r[1] = segs[static_cast<size_t>(s)].z_hi;
                    r[2] = segs[static_cast<size_t>(s)].dz;
                }
            }
        }

        if (stats) {
            stats[0] = S;
            stats[1] = covered;
            stats[2] = marked;
        }
        if (timing_ms) {
            timing_ms[0] = t1 - t0;
            timing_ms[1] = t2 - t1;
            timing_ms[2] = t3 - t2;
        }
        return 0;
    } catch (const std::exception& e) {
        set_err(err, errlen, e.what());
        return 1;
    }
}

const char* c5o_kind() { return "port"; }

}  // extern "C"
