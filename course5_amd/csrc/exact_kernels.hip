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

// SoA in -> SoA out (volume grid vertices): one thread per vertex, fully coalesced.
__global__ __launch_bounds__(256) void transform_points_soa(const double* __restrict__ px,
                                                            const double* __restrict__ py,
                                                            const double* __restrict__ pz,
                                                            double* __restrict__ vx,
                                                            double* __restrict__ vy,
                                                            double* __restrict__ vz, int64_t n,
                                                            RotationList R) {
    const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
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

// One thread per UNIQUE solid face (host: unique_solid_faces).  pts: transformed points [m][3].
// mask[local pixel] receives the largest (slot + 1) of the solid objects covering it: objects
// later in the tetra vector overwrite earlier ones in the serial reference (line.cpp:246-249), and
// all cells of one object share a colour.
__global__ __launch_bounds__(256) void solid_mask_raster(const double* __restrict__ pts,
                                                         const int4* __restrict__ faces, int64_t n_faces,
                                                         uint32_t value, const double* __restrict__ Ytab,
                                                         ImageParams im, uint32_t* __restrict__ mask) {
    const int64_t gid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (gid >= n_faces) return;
    const int4 fc = faces[gid];
    // three projected points, kept in registers (runtime-indexed arrays would live in scratch)
    double ax = pts[3 * static_cast<size_t>(fc.x)], ay = pts[3 * static_cast<size_t>(fc.x) + 1];
    double bx = pts[3 * static_cast<size_t>(fc.y)], by = pts[3 * static_cast<size_t>(fc.y) + 1];
    double cx = pts[3 * static_cast<size_t>(fc.z)], cy = pts[3 * static_cast<size_t>(fc.z) + 1];
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
    const double p0[2] = {ax, ay}, p1[2] = {bx, by}, p2[2] = {cx, cy};

    // plane.cpp:66-89
    const double side = edge_side(p0, p2, p1);
    const bool up_left = (p0[0] >= p2[0]) && (side >= 0);
    const bool down_right = (p0[0] < p2[0]) && (side > 0);
    const bool long_edge_is_left = !(up_left || down_right);

    // plane.cpp:96-97 (double -> size_t conversions of non-negative values)
    const long long row_hi = static_cast<long long>(floor(frac_y(im, p0[1])));
    const long long row_lo = static_cast<long long>(ceil(frac_y(im, p2[1])));
    if (row_hi < row_lo) return;

    // per-face constants of the division-free path: x(y) = (ax - bx) * (y - ay) * inv(ay - by) + ax
    const double hi_col = static_cast<double>(im.res_x) - 1;
    const double inv_step_x = 1.0 / im.step_x;
    const double d02 = p0[1] - p2[1], d21 = p2[1] - p1[1], d01 = p0[1] - p1[1];
    const bool flat02 = fabs(d02) < DBL_EPSILON, flat21 = fabs(d21) < DBL_EPSILON, flat01 = fabs(d01) < DBL_EPSILON;
    const double i02 = flat02 ? 0.0 : 1.0 / d02, i21 = flat21 ? 0.0 : 1.0 / d21, i01 = flat01 ? 0.0 : 1.0 / d01;
    const double s02 = p0[0] - p2[0], s21 = p2[0] - p1[0], s01 = p0[0] - p1[0];

    for (long long row = row_lo; row <= row_hi; ++row) {
        const int lrow = local_row_of(im, static_cast<int>(row));
        if (lrow < 0) continue;
        const double y = Ytab[row];  // == _lines[0][row_lo].y() + k * step_y accumulated (plane.cpp:100,138)
        const bool below_mid = y < p1[1];
        // plane.cpp:106-122 without divisions
        const double xl_fast = flat02 ? p0[0] : s02 * (y - p0[1]) * i02 + p0[0];
        const double xs_fast = below_mid ? (flat21 ? p2[0] : s21 * (y - p2[1]) * i21 + p2[0])
                                         : (flat01 ? p0[0] : s01 * (y - p0[1]) * i01 + p0[0]);
        auto x_long = [&]() { return edge_x_at(p0, p2, y); };
        auto x_short = [&]() { return edge_x_at(below_mid ? p2 : p0, p1, y); };
        long long col_lo, col_hi;
        if (long_edge_is_left) {
            col_lo = column_of<true>(xl_fast, im, inv_step_x, hi_col, x_long);
            col_hi = column_of<false>(xs_fast, im, inv_step_x, hi_col, x_short);
        } else {
            col_lo = column_of<true>(xs_fast, im, inv_step_x, hi_col, x_short);
            col_hi = column_of<false>(xl_fast, im, inv_step_x, hi_col, x_long);
        }
        uint32_t* mrow = mask + static_cast<size_t>(lrow) * im.res_x;
        // Unconditional plain store: one launch handles one slot, so every thread of the launch writes
        // the same value (a benign race), and launches of higher slots run later on the same stream and
        // overwrite, like later cells in the serial reference.  No read-before-write: a dependent load
        // per pixel in this serial loop cost 3.4x (0.55 vs 0.16 ms for the Roche lobe at 2400x1800),
        // atomics 7x.
        for (long long col = col_lo; col <= col_hi; ++col) mrow[col] = value;
    }
}

void launch_transform_soa(hipStream_t s, const double* px, const double* py, const double* pz,
                          double* vx, double* vy, double* vz, int64_t n, const RotationList& R) {
    if (n <= 0) return;
    const unsigned blocks = static_cast<unsigned>((n + 255) / 256);
    hipLaunchKernelGGL(transform_points_soa, dim3(blocks), dim3(256), 0, s, px, py, pz, vx, vy, vz, n, R);
}

void launch_transform_aos(hipStream_t s, const double* in, double* out, int64_t n,
                          const RotationList& R) {
    if (n <= 0) return;
    const unsigned blocks = static_cast<unsigned>((n + 255) / 256);
    hipLaunchKernelGGL(transform_points_aos, dim3(blocks), dim3(256), 0, s, in, out, n, R);
}

void launch_solid_mask_raster(hipStream_t s, const double* pts, const int4* faces, int64_t n_faces,
                              uint32_t value, const double* Ytab, const ImageParams& im, uint32_t* mask) {
    if (n_faces <= 0) return;
    const unsigned blocks = static_cast<unsigned>((n_faces + 255) / 256);
    hipLaunchKernelGGL(solid_mask_raster, dim3(blocks), dim3(256), 0, s, pts, faces, n_faces, value, Ytab, im, mask);
}

}  // namespace c5
