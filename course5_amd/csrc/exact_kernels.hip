// Kernels whose results must equal the reference's host arithmetic bit for bit.
// This translation unit is compiled with -ffp-contract=off: every multiply and add rounds
// separately, as in the reference's -O3 x86-64 build (no FMA contraction there).
//
//   transform_points_*   object3d_base::rotate_around_{x,y}_axis (object3d_base.cpp:202-219)
//                        -> tetra::point_rotate_around_* (tetra.cpp:44-62), three sequential
//                        in-place rotations (main.cpp:105-107); cos/sin come from the host.
//   solid_mask_raster    plane::find_intersections_with_polygon for solid cells
//                        (plane.cpp:57-142, is_solid branch :130-131 -> line.cpp:246-249),
//                        including the clamped pixel mapping (plane.cpp:194-212).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>

#include "device_types.hpp"
#include "kernels.hpp"

namespace c5 {

__device__ __forceinline__ void rotate_point(const RotationList& R, double& x, double& y, double& z) {
    for (int r = 0; r < R.n; ++r) {
        const double c = R.cosv[r], s = R.sinv[r];
        if (R.axis[r] == 0) {  // tetra.cpp:44-48
            const double y_old = y;
            y = y * c - z * s;
            z = y_old * s + z * c;
        } else {  // tetra.cpp:51-62
            x -= R.x0[r];
            const double x_old = x;
            x = x * c - z * s;
            z = x_old * s + z * c;
            x += R.x0[r];
        }
    }
}

constexpr int64_t kCountersDwords = static_cast<int64_t>(sizeof(FrameCounters)) * kCounterLines / 4;  // (the shards + DepthSamples)

// SoA in -> SoA out (volume grid vertices): one thread per vertex, fully coalesced.
__global__ __launch_bounds__(256) void transform_points_soa(const double* __restrict__ px,
                                                            const double* __restrict__ py,
                                                            const double* __restrict__ pz,
                                                            double* __restrict__ vx,
                                                            double* __restrict__ vy,
                                                            double* __restrict__ vz, int64_t n,
                                                            RotationList R, uint32_t* __restrict__ counters,
                                                            uint32_t* __restrict__ sb, int n_sb) {
    const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (sb && i < n_sb) sb[i] = 0u;  // the walk's per-row costs of the frame before (read by the host when it waited)
    // first kernel of a frame: it also clears the frame's statistics (one API call less per frame than a
    // memset of its own); the grid covers at least kCountersDwords threads
    if (counters && i < kCountersDwords) counters[i] = 0u;
    if (i >= n) return;
    double x = px[i], y = py[i], z = pz[i];
    rotate_point(R, x, y, z);
    vx[i] = x;
    vy[i] = y;
    vz[i] = z;
}

// AoS xyz rows (solid tet soups: [n_tets][4][3]).
__global__ __launch_bounds__(256) void transform_points_aos(const double* __restrict__ in,
                                                            double* __restrict__ out, int64_t n,
                                                            RotationList R) {
    const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (i >= n) return;
    double x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
    rotate_point(R, x, y, z);
    out[3 * i] = x;
    out[3 * i + 1] = y;
    out[3 * i + 2] = z;
}

// plane.cpp:194-212
__device__ __forceinline__ double frac_x(const ImageParams& im, double x) {
    const double r = (x - im.x_min) / im.step_x;
    const double hi = static_cast<double>(im.res_x) - 1;
    if (r < 0) return 0;
    if (r > hi) return hi;
    return r;
}
__device__ __forceinline__ double frac_y(const ImageParams& im, double y) {
    const double r = (y - im.y_min) / im.step_y;
    const double hi = static_cast<double>(im.res_y) - 1;
    if (r < 0) return 0;
    if (r > hi) return hi;
    return r;
}
// plane.cpp:46-48
__device__ __forceinline__ double edge_side(const double* a, const double* b, const double* p) {
    return (b[1] - a[1]) * p[0] + (a[0] - b[0]) * p[1] + (b[0] * a[1] - a[0] * b[1]);
}
// plane.cpp:50-55
__device__ __forceinline__ double edge_x_at(const double* a, const double* b, double y) {
    if (fabs(a[1] - b[1]) < DBL_EPSILON) return a[0];
    return (a[0] - b[0]) * (y - a[1]) / (a[1] - b[1]) + a[0];
}

// Pixel column of x under the reference's clamped mapping (plane.cpp:194-202) followed by
// floor (UP = false) or ceil (UP = true), evaluated without a division when that is provably the
// same: x and 1/step are accurate to a few ulp, so unless the quotient lies within 1e-7 of an
// integer (or of a clamp bound) the rounded index cannot differ from the exactly divided one.
// Otherwise `exact` is called and the reference's own operation sequence decides.
template <bool UP, class Exact>
__device__ __forceinline__ long long column_of(double x_fast, const ImageParams& im, double inv_step_x, double hi,
                                               Exact&& exact) {
    const double r = (x_fast - im.x_min) * inv_step_x;
    const double n = rint(r);
    if (!(fabs(r - n) > 1e-7) || !(r > 1e-7) || !(r < hi - 1e-7)) {
        if (r < -1.0) return 0;          // far below the lower clamp: 0 either way
        if (r > hi + 1.0) return static_cast<long long>(hi);  // far above the upper clamp
        const double re = frac_x(im, exact());
        return static_cast<long long>(UP ? ceil(re) : floor(re));
    }
    return static_cast<long long>(UP ? ceil(r) : floor(r));
}

// Scan conversion of one projected triangle under the reference's inclusive rule
// (plane::find_intersections_with_polygon, plane.cpp:57-142), set up once per face and queried per
// image row.  Shared by the solid mask raster and by the bin-sort-resolve path.
struct FaceScan {
    double p0[2], p1[2], p2[2];  // vertices by descending y (plane.cpp:61)
    bool long_edge_is_left;
    long long row_lo, row_hi;    // plane.cpp:96-97
    // constants of the division-free evaluation
    double hi_col, inv_step_x;
    bool flat02, flat21, flat01;
    double i02, i21, i01, s02, s21, s01;

    __device__ __forceinline__ void setup(double ax, double ay, double bx, double by, double cx, double cy,
                                          const ImageParams& im) {
        // plane.cpp:61 — std::sort of three pointers by descending y == stable insertion sort.  (The
        // input order only matters when two y are equal: a degenerate face.)
        if (by > ay) {  // insert b before a
            double t = ax; ax = bx; bx = t;
            t = ay; ay = by; by = t;
        }
        if (cy > ay) {  // c goes to the front: (c, a, b)
            const double tx = cx, ty = cy;
            cx = bx; cy = by;
            bx = ax; by = ay;
            ax = tx; ay = ty;
        } else if (cy > by) {  // (a, c, b)
            double t = bx; bx = cx; cx = t;
            t = by; by = cy; cy = t;
        }
        p0[0] = ax; p0[1] = ay;
        p1[0] = bx; p1[1] = by;
        p2[0] = cx; p2[1] = cy;
        // plane.cpp:66-89
        const double side = edge_side(p0, p2, p1);
        const bool up_left = (p0[0] >= p2[0]) && (side >= 0);
        const bool down_right = (p0[0] < p2[0]) && (side > 0);
        long_edge_is_left = !(up_left || down_right);
        // plane.cpp:96-97 (double -> size_t conversions of non-negative values)
        row_hi = static_cast<long long>(floor(frac_y(im, p0[1])));
        row_lo = static_cast<long long>(ceil(frac_y(im, p2[1])));
        // x(y) = (ax - bx) * (y - ay) * inv(ay - by) + ax
        hi_col = static_cast<double>(im.res_x) - 1;
        inv_step_x = 1.0 / im.step_x;
        const double d02 = p0[1] - p2[1], d21 = p2[1] - p1[1], d01 = p0[1] - p1[1];
        flat02 = fabs(d02) < DBL_EPSILON;
        flat21 = fabs(d21) < DBL_EPSILON;
        flat01 = fabs(d01) < DBL_EPSILON;
        i02 = flat02 ? 0.0 : 1.0 / d02;
        i21 = flat21 ? 0.0 : 1.0 / d21;
        i01 = flat01 ? 0.0 : 1.0 / d01;
        s02 = p0[0] - p2[0];
        s21 = p2[0] - p1[0];
        s01 = p0[0] - p1[0];
    }

    // columns covered in the row whose ray has coordinate y (plane.cpp:106-127); empty if lo > hi
    __device__ __forceinline__ void span(double y, const ImageParams& im, long long& col_lo, long long& col_hi) const {
        const bool below_mid = y < p1[1];
        const double xl_fast = flat02 ? p0[0] : s02 * (y - p0[1]) * i02 + p0[0];
        const double xs_fast = below_mid ? (flat21 ? p2[0] : s21 * (y - p2[1]) * i21 + p2[0])
                                         : (flat01 ? p0[0] : s01 * (y - p0[1]) * i01 + p0[0]);
        auto x_long = [&]() { return edge_x_at(p0, p2, y); };
        auto x_short = [&]() { return edge_x_at(below_mid ? p2 : p0, p1, y); };
        if (long_edge_is_left) {
            col_lo = column_of<true>(xl_fast, im, inv_step_x, hi_col, x_long);
            col_hi = column_of<false>(xs_fast, im, inv_step_x, hi_col, x_short);
        } else {
            col_lo = column_of<true>(xs_fast, im, inv_step_x, hi_col, x_short);
            col_hi = column_of<false>(xl_fast, im, inv_step_x, hi_col, x_long);
        }
    }
};

// LANES threads per UNIQUE solid face (host: unique_solid_faces), rows of the face dealt among them: the
// kernel's time is the longest chain of rows one thread walks (the centre-fan slivers of the Roche lobe
// are hundreds of rows tall at 2400x1800), not its total work.  pts: transformed points [m][3].
// mask[local pixel] receives the largest (slot + 1) of the solid objects covering it: objects
// later in the tetra vector overwrite earlier ones in the serial reference (line.cpp:246-249), and
// all cells of one object share a colour.
__global__ __launch_bounds__(256) void solid_mask_raster(const double* __restrict__ pts,
                                                         const int4* __restrict__ faces, int64_t n_faces,
                                                         uint32_t value, const double* __restrict__ Ytab,
                                                         ImageParams im, uint32_t* __restrict__ mask, int lanes_log2,
                                                         int only_across_border) {
    const int64_t tid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    const int64_t gid = tid >> lanes_log2;
    const long long sub = static_cast<long long>(tid & ((1ll << lanes_log2) - 1)), n_sub = 1ll << lanes_log2;
    if (gid >= n_faces) return;
    const int4 fc = faces[gid];
    const double ax = pts[3 * static_cast<size_t>(fc.x)], ay = pts[3 * static_cast<size_t>(fc.x) + 1];
    const double bx = pts[3 * static_cast<size_t>(fc.y)], by = pts[3 * static_cast<size_t>(fc.y) + 1];
    const double cx = pts[3 * static_cast<size_t>(fc.z)], cy = pts[3 * static_cast<size_t>(fc.z) + 1];
    if (only_across_border) {
        // an INTERIOR face of its solid (c_api.hip: enqueue_solids): it covers nothing the solid's other faces do not
        // cover - unless get_pixel_by_x/_y's clamp (plane.cpp:194-212) has a hand in its pixels; a face that lies a
        // pixel inside the domain on every side is left out
        const double x_a = im.x_min, x_b = im.x_min + im.step_x * (im.res_x - 1);
        const double y_a = im.y_min, y_b = im.y_min + im.step_y * (im.res_y - 1);
        const double mx = 2.0 * fabs(im.step_x), my = 2.0 * fabs(im.step_y);
        if (fmin(ax, fmin(bx, cx)) > fmin(x_a, x_b) + mx && fmax(ax, fmax(bx, cx)) < fmax(x_a, x_b) - mx &&
            fmin(ay, fmin(by, cy)) > fmin(y_a, y_b) + my && fmax(ay, fmax(by, cy)) < fmax(y_a, y_b) - my)
            return;
    }
    FaceScan fs;
    fs.setup(ax, ay, bx, by, cx, cy, im);
    for (long long row = fs.row_lo + sub; row <= fs.row_hi; row += n_sub) {
        const int lrow = local_row_of(im, static_cast<int>(row));
        if (lrow < 0) continue;
        const double y = Ytab[row];  // == _lines[0][row_lo].y() + k * step_y accumulated (plane.cpp:100,138)
        long long col_lo, col_hi;
        fs.span(y, im, col_lo, col_hi);
        uint32_t* mrow = mask + static_cast<size_t>(lrow) * im.res_x;
        // Unconditional plain store: one launch handles one slot, so every thread of the launch writes
        // the same value (a benign race), and launches of higher slots run later on the same stream and
        // overwrite, like later cells in the serial reference.  No read-before-write: a dependent load
        // per pixel in this serial loop cost 3.4x (0.55 vs 0.16 ms for the Roche lobe at 2400x1800),
        // atomics 7x.
        for (long long col = col_lo; col <= col_hi; ++col) mrow[col] = value;
    }
}

// ------------------------------------------------------------------------------------------
// bin_sort_resolve: the reference's own algorithm on the GPU, for grids a face-adjacency walk
// cannot handle (tet soups, overlapping or non-conforming cells) and as a second, independent
// implementation to check the walk against at full size.
//   bin_cells<0/1>   plane::find_intersections (plane.cpp:184-192,14-44): every face of every cell is
//                    scan-converted; a pixel covered by two faces of a cell gets one (z_hi, dz, cell)
//                    record (line.cpp:29-67 pairing, line.cpp:99-131,150-174 z of both faces).
//                    Pass 0 counts per pixel, pass 1 fills the CSR lists.
//   resolve_pixels   plane::trace_rays (plane.cpp:161-169): per pixel sort by z_hi descending
//                    (line.cpp:138), tau front to back (line.cpp:176-193), I back to front with the
//                    reference's recurrence and a true division (line.cpp:195-227).
// One wavefront per cell, lanes over the rows of its bounding box.
// ------------------------------------------------------------------------------------------
struct alignas(8) Segment {
    double z_hi;
    double dz;
    long long cell;
};

// line.cpp:150-174 with the reference's operation order
__device__ __forceinline__ double face_z_reference(double x, double y, const double* a, const double* b, const double* c) {
    const double tx = (x - a[0]) * ((b[1] - a[1]) * (c[2] - a[2]) - (c[1] - a[1]) * (b[2] - a[2]));
    const double ty = (y - a[1]) * ((b[0] - a[0]) * (c[2] - a[2]) - (c[0] - a[0]) * (b[2] - a[2]));
    const double den = ((b[0] - a[0]) * (c[1] - a[1]) - (c[0] - a[0]) * (b[1] - a[1]));
    return (ty - tx) / den + a[2];
}

template <int PASS>
__global__ __launch_bounds__(256) void bin_cells(GridView g, const double* __restrict__ Xtab,
                                                 const double* __restrict__ Ytab, ImageParams im,
                                                 int32_t* __restrict__ count, const int64_t* __restrict__ offs,
                                                 Segment* __restrict__ segs, unsigned* __restrict__ odd_pixels) {
    const int lane = threadIdx.x & 63;
    const int64_t cell = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (cell >= g.n_cells) return;
    const int4 cv = g.cell_vert[cell];
    double p[4][3];
    {
        const int vid[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            p[k][0] = g.vx[vid[k]];
            p[k][1] = g.vy[vid[k]];
            p[k][2] = g.vz[vid[k]];
        }
    }
    // faces 0:(0,1,2) 1:(0,1,3) 2:(0,2,3) 3:(1,2,3) (plane.cpp:30-37)
    FaceScan f0, f1, f2, f3;
    f0.setup(p[0][0], p[0][1], p[1][0], p[1][1], p[2][0], p[2][1], im);
    f1.setup(p[0][0], p[0][1], p[1][0], p[1][1], p[3][0], p[3][1], im);
    f2.setup(p[0][0], p[0][1], p[2][0], p[2][1], p[3][0], p[3][1], im);
    f3.setup(p[1][0], p[1][1], p[2][0], p[2][1], p[3][0], p[3][1], im);
    const long long r_lo = min(min(f0.row_lo, f1.row_lo), min(f2.row_lo, f3.row_lo));
    const long long r_hi = max(max(f0.row_hi, f1.row_hi), max(f2.row_hi, f3.row_hi));

    for (long long row = r_lo + lane; row <= r_hi; row += 64) {
        const int lrow = local_row_of(im, static_cast<int>(row));
        if (lrow < 0) continue;
        const double y = Ytab[row];
        long long lo0 = 1, hi0 = 0, lo1 = 1, hi1 = 0, lo2 = 1, hi2 = 0, lo3 = 1, hi3 = 0;
        if (row >= f0.row_lo && row <= f0.row_hi) f0.span(y, im, lo0, hi0);
        if (row >= f1.row_lo && row <= f1.row_hi) f1.span(y, im, lo1, hi1);
        if (row >= f2.row_lo && row <= f2.row_hi) f2.span(y, im, lo2, hi2);
        if (row >= f3.row_lo && row <= f3.row_hi) f3.span(y, im, lo3, hi3);
        long long c_lo = im.res_x, c_hi = -1;
        if (lo0 <= hi0) { c_lo = min(c_lo, lo0); c_hi = max(c_hi, hi0); }
        if (lo1 <= hi1) { c_lo = min(c_lo, lo1); c_hi = max(c_hi, hi1); }
        if (lo2 <= hi2) { c_lo = min(c_lo, lo2); c_hi = max(c_hi, hi2); }
        if (lo3 <= hi3) { c_lo = min(c_lo, lo3); c_hi = max(c_hi, hi3); }
        for (long long col = c_lo; col <= c_hi; ++col) {
            const unsigned m = (col >= lo0 && col <= hi0 ? 1u : 0u) | (col >= lo1 && col <= hi1 ? 2u : 0u) |
                               (col >= lo2 && col <= hi2 ? 4u : 0u) | (col >= lo3 && col <= hi3 ? 8u : 0u);
            const int hits = __popc(m);
            if (hits == 0) continue;
            if (hits & 1) {  // the reference mis-pairs or aborts here (plane.cpp:39-41, line.cpp:40-47)
                if (PASS == 0) atomicAdd(odd_pixels, 1u);
                continue;
            }
            const size_t lp = static_cast<size_t>(lrow) * im.res_x + static_cast<size_t>(col);
            const int pairs = hits >> 1;  // faces pair up in scan order: (1st, 2nd), (3rd, 4th) (line.cpp:49-54)
            if (PASS == 0) {
                atomicAdd(count + lp, pairs);
            } else {
                const int k = atomicSub(count + lp, pairs) - pairs;
                const double x = Xtab[col];
                double zf[4];
                zf[0] = (m & 1u) ? face_z_reference(x, y, p[0], p[1], p[2]) : 0.0;
                zf[1] = (m & 2u) ? face_z_reference(x, y, p[0], p[1], p[3]) : 0.0;
                zf[2] = (m & 4u) ? face_z_reference(x, y, p[0], p[2], p[3]) : 0.0;
                zf[3] = (m & 8u) ? face_z_reference(x, y, p[1], p[2], p[3]) : 0.0;
                // first pair = the two lowest-numbered covering faces, second pair (if any) the others
                double za = 0, zb = 0, zc = 0, zd = 0;
                int seen = 0;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    if (m & (1u << f)) {
                        if (seen == 0) za = zf[f];
                        else if (seen == 1) zb = zf[f];
                        else if (seen == 2) zc = zf[f];
                        else zd = zf[f];
                        ++seen;
                    }
                }
                Segment sgm;
                sgm.cell = cell;
                sgm.z_hi = fmax(za, zb);  // line.cpp:124-133
                sgm.dz = sgm.z_hi - fmin(za, zb);
                segs[offs[lp] + k] = sgm;
                if (pairs == 2) {
                    sgm.z_hi = fmax(zc, zd);
                    sgm.dz = sgm.z_hi - fmin(zc, zd);
                    segs[offs[lp] + k + 1] = sgm;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void resolve_pixels(GridView g, ImageParams im, const int32_t* __restrict__ count0,
                                                      const int64_t* __restrict__ offs, Segment* __restrict__ segs,
                                                      const uint32_t* __restrict__ mask, SolidTable solids,
                                                      double alpha_limit, float2* __restrict__ out,
                                                      FrameCounters* counters) {
    const int64_t lp = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    const int64_t n_px = static_cast<int64_t>(im.n_local_rows) * im.res_x;
    unsigned covered = 0, solid = 0;
    if (lp < n_px) {
        float2 result = make_float2(0.f, 0.f);
        const uint32_t mv = mask ? mask[lp] : 0u;
        Segment* list = segs + offs[lp];
        const int n = static_cast<int>(offs[lp + 1] - offs[lp]);
        if (mv) {  // line.cpp:177-179,197-199
            double colour = 0.0;
            for (int s = 0; s < solids.n_slots; ++s)
                if (mv == static_cast<uint32_t>(s) + 1u) colour = solids.colour[s];
            result.x = static_cast<float>(colour);
            result.y = result.x;
            solid = 1;
        } else if (n > 0) {
            covered = 1;
            // line.cpp:138: descending z_hi (Shell sort in place; ties are unordered in the reference too)
            for (int gap = n / 2; gap > 0; gap = (gap == 2) ? 1 : static_cast<int>(gap / 2.2)) {
                for (int i = gap; i < n; ++i) {
                    const Segment t = list[i];
                    int j = i;
                    while (j >= gap && list[j - gap].z_hi < t.z_hi) {
                        list[j] = list[j - gap];
                        j -= gap;
                    }
                    list[j] = t;
                }
            }
            double sum = 0.0;  // line.cpp:182-190
            for (int i = 0; i < n; ++i) sum = sum + list[i].dz * g.alpha[list[i].cell];
            double I = 0.0;  // line.cpp:201-225
            for (int i = n - 1; i >= 0; --i) {
                const double Q = g.q[list[i].cell];
                double a = g.alpha[list[i].cell];
                if (a > alpha_limit) a = alpha_limit;
                const double C = Q - a * I;
                if (!(a < DBL_EPSILON)) I = (Q - C * exp(-a * list[i].dz)) / a;
            }
            result.x = static_cast<float>(sum);
            result.y = static_cast<float>(I);
        }
        out[lp] = result;
    }
    // statistics
    for (int d = 32; d >= 1; d >>= 1) {
        covered += __shfl_xor(covered, d);
        solid += __shfl_xor(solid, d);
    }
    if ((threadIdx.x & 63) == 0) {
        if (covered) atomicAdd(&counters->steps_cov, static_cast<unsigned long long>(covered) << kCounterHighShift);
        if (solid) atomicAdd(&counters->ent_solid, static_cast<unsigned long long>(solid) << kCounterHighShift);
    }
    (void)count0;
}

// exclusive scan of int32 counts into int64 offsets (segment totals exceed 2^31 at large sizes):
// single-block-per-chunk three-step scan, 1024 items per block
__global__ __launch_bounds__(256) void scan64_block_sums(const int32_t* __restrict__ count, int64_t n,
                                                         int64_t* __restrict__ sums) {
    __shared__ long long part[256];
    const int64_t base = (blockIdx.x * 256ll + threadIdx.x) * 4;
    long long v = 0;
    for (int k = 0; k < 4; ++k)
        if (base + k < n) v += count[base + k];
    part[threadIdx.x] = v;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (static_cast<int>(threadIdx.x) < d) part[threadIdx.x] += part[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = part[0];
}

__global__ void scan64_sums_serial(int64_t* sums, int64_t n_blocks) {  // n_blocks is a few thousand
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    long long run = 0;
    for (int64_t i = 0; i < n_blocks; ++i) {
        const long long v = sums[i];
        sums[i] = run;
        run += v;
    }
    sums[n_blocks] = run;
}

__global__ __launch_bounds__(256) void scan64_finish(const int32_t* __restrict__ count, int64_t n,
                                                     const int64_t* __restrict__ sums, int64_t* __restrict__ offs,
                                                     int64_t n_blocks) {
    __shared__ long long part[256];
    const int64_t base = (blockIdx.x * 256ll + threadIdx.x) * 4;
    int c[4];
    long long v = 0;
    for (int k = 0; k < 4; ++k) {
        c[k] = (base + k < n) ? count[base + k] : 0;
        v += c[k];
    }
    part[threadIdx.x] = v;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 256 partial sums
    for (int d = 1; d < 256; d <<= 1) {
        const long long add = (static_cast<int>(threadIdx.x) >= d) ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    long long run = sums[blockIdx.x] + part[threadIdx.x] - v;
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) offs[base + k] = run;
        run += c[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) offs[n] = sums[n_blocks];
}

void launch_bin_count(hipStream_t s, const GridView& g, const double* Xtab, const double* Ytab,
                      const ImageParams& im, int32_t* count, unsigned* odd_pixels) {
    if (g.n_cells <= 0) return;
    const unsigned blocks = static_cast<unsigned>((g.n_cells + 3) / 4);
    hipLaunchKernelGGL(bin_cells<0>, dim3(blocks), dim3(256), 0, s, g, Xtab, Ytab, im, count,
                       static_cast<const int64_t*>(nullptr), static_cast<Segment*>(nullptr), odd_pixels);
}

void launch_bin_fill(hipStream_t s, const GridView& g, const double* Xtab, const double* Ytab,
                     const ImageParams& im, int32_t* count, const int64_t* offs, void* segs) {
    if (g.n_cells <= 0) return;
    const unsigned blocks = static_cast<unsigned>((g.n_cells + 3) / 4);
    hipLaunchKernelGGL(bin_cells<1>, dim3(blocks), dim3(256), 0, s, g, Xtab, Ytab, im, count, offs,
                       static_cast<Segment*>(segs), static_cast<unsigned*>(nullptr));
}

void launch_scan64(hipStream_t s, const int32_t* count, int64_t* offs, int64_t n, int64_t* scratch) {
    if (n <= 0) return;
    const int64_t n_blocks = (n + 1023) / 1024;
    hipLaunchKernelGGL(scan64_block_sums, dim3(static_cast<unsigned>(n_blocks)), dim3(256), 0, s, count, n, scratch);
    hipLaunchKernelGGL(scan64_sums_serial, dim3(1), dim3(64), 0, s, scratch, n_blocks);
    hipLaunchKernelGGL(scan64_finish, dim3(static_cast<unsigned>(n_blocks)), dim3(256), 0, s, count, n, scratch, offs,
                       n_blocks);
}

void launch_resolve(hipStream_t s, const GridView& g, const ImageParams& im, const int64_t* offs, void* segs,
                    const uint32_t* mask, const SolidTable& solids, double alpha_limit, float2* out,
                    FrameCounters* counters) {
    const int64_t n_px = static_cast<int64_t>(im.n_local_rows) * im.res_x;
    if (n_px <= 0) return;
    const unsigned blocks = static_cast<unsigned>((n_px + 255) / 256);
    hipLaunchKernelGGL(resolve_pixels, dim3(blocks), dim3(256), 0, s, g, im, static_cast<const int32_t*>(nullptr), offs,
                       static_cast<Segment*>(segs), mask, solids, alpha_limit, out, counters);
}

size_t segment_bytes() { return sizeof(Segment); }

// `raster_from`: the counters of the frame whose raster built the entry lists this frame reuses, if they are not
// `counters` themselves (frames delivered to host memory keep their statistics per frame): its share is copied over.
__global__ __launch_bounds__(128) void clear_walk_counters(FrameCounters* __restrict__ counters, uint32_t* __restrict__ sb, int n_sb,
                                                           const FrameCounters* __restrict__ raster_from) {
    const int i = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
    if (sb && i < n_sb) sb[i] = 0u;
    if (i < kCounterShards) {
        FrameCounters& c = counters[i];
        c.seg_tiles = 0;
        c.steps_cov = 0;
        c.ent_solid = 0;
        c.walk_overflow = 0;
        c.overlap_rays = 0;
        c.odd_pixels = 0;
        c.seg_max = 0;
        c.exit_max_key = 0;
        c.entry_min_key = 0;
        if (raster_from) {
            c.entry_overflow = raster_from[i].entry_overflow;
            c.pool_used = raster_from[i].pool_used;
        }
    }
    if (i < kFitSlots) {  // DepthSamples: the walk's start anew; the raster's stay (or come over with its other counters)
        DepthSamples* fit = reinterpret_cast<DepthSamples*>(counters + kCounterShards);
        fit->exit_key[i] = 0ull;
        if (raster_from) fit->entry_key[i] = reinterpret_cast<const DepthSamples*>(raster_from + kCounterShards)->entry_key[i];
    }
}

void launch_clear_walk_counters(hipStream_t s, FrameCounters* counters, uint32_t* sb, int n_sb, const FrameCounters* raster_from) {
    const int n = std::max(std::max(kCounterShards, kFitSlots), sb ? n_sb : 0);
    hipLaunchKernelGGL(clear_walk_counters, dim3(static_cast<unsigned>((n + 127) / 128)), dim3(128), 0, s, counters, sb, n_sb,
                       raster_from == counters ? nullptr : raster_from);
}

void launch_transform_soa(hipStream_t s, const double* px, const double* py, const double* pz,
                          double* vx, double* vy, double* vz, int64_t n, const RotationList& R,
                          FrameCounters* counters_to_clear, uint32_t* sb, int n_sb) {
    int64_t threads = counters_to_clear ? (n > kCountersDwords ? n : kCountersDwords) : n;
    if (sb && threads < 256) threads = 256;
    if (threads <= 0) return;
    const unsigned blocks = static_cast<unsigned>((threads + 255) / 256);
    hipLaunchKernelGGL(transform_points_soa, dim3(blocks), dim3(256), 0, s, px, py, pz, vx, vy, vz, n, R,
                       reinterpret_cast<uint32_t*>(counters_to_clear), sb, n_sb);
}

void launch_transform_aos(hipStream_t s, const double* in, double* out, int64_t n,
                          const RotationList& R) {
    if (n <= 0) return;
    const unsigned blocks = static_cast<unsigned>((n + 255) / 256);
    hipLaunchKernelGGL(transform_points_aos, dim3(blocks), dim3(256), 0, s, in, out, n, R);
}

__global__ __launch_bounds__(256) void mask_overlay(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n4) {
    const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (i >= n4) return;
    const uint4 v = src[i];
    if ((v.x | v.y | v.z | v.w) == 0u) return;  // most of the image: nothing to lay over
    uint4 d = dst[i];
    d.x = v.x ? v.x : d.x;
    d.y = v.y ? v.y : d.y;
    d.z = v.z ? v.z : d.z;
    d.w = v.w ? v.w : d.w;
    dst[i] = d;
}

void launch_mask_overlay(hipStream_t s, const uint32_t* src, uint32_t* dst, int64_t n_padded) {
    const int64_t n4 = n_padded / 4;  // masks are padded to 1024 entries
    if (n4 <= 0) return;
    hipLaunchKernelGGL(mask_overlay, dim3(static_cast<unsigned>((n4 + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), n4);
}

void launch_solid_mask_raster(hipStream_t s, const double* pts, const int4* faces, int64_t n_faces,
                              uint32_t value, const double* Ytab, const ImageParams& im, uint32_t* mask,
                              int lanes_per_face, bool only_across_border) {
    if (n_faces <= 0) return;
    int lanes_log2 = 0;
    while ((2 << lanes_log2) <= lanes_per_face && lanes_log2 < 6) ++lanes_log2;
    const int64_t threads = n_faces << lanes_log2;
    const unsigned blocks = static_cast<unsigned>((threads + 255) / 256);
    hipLaunchKernelGGL(solid_mask_raster, dim3(blocks), dim3(256), 0, s, pts, faces, n_faces, value, Ytab, im, mask,
                       lanes_log2, only_across_border ? 1 : 0);
}

}  // namespace c5
