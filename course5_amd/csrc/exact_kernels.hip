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

// One thread per solid face.  tets: transformed [n][4][3].  mask[local pixel] receives the
// largest (solid id + 1) covering it: the cell the serial reference would have written last.
__global__ __launch_bounds__(256) void solid_mask_raster(const double* __restrict__ tets,
                                                         int64_t n_tets, uint32_t first_id,
                                                         const double* __restrict__ Ytab,
                                                         ImageParams im,
                                                         uint32_t* __restrict__ mask) {
    const int64_t gid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (gid >= 4 * n_tets) return;
    const int64_t tet = gid >> 2;
    const int f = static_cast<int>(gid & 3);
    // face -> vertices (plane.cpp:30-37)
    const int i0 = (f == 3) ? 1 : 0;
    const int i1 = (f <= 1) ? 1 : 2;
    const int i2 = (f == 0) ? 2 : 3;
    const double* base = tets + 12 * tet;
    double v[3][2];
    v[0][0] = base[3 * i0];
    v[0][1] = base[3 * i0 + 1];
    v[1][0] = base[3 * i1];
    v[1][1] = base[3 * i1 + 1];
    v[2][0] = base[3 * i2];
    v[2][1] = base[3 * i2 + 1];
    // plane.cpp:61 — std::sort of three pointers by descending y == stable insertion sort
    int o0 = 0, o1 = 1, o2 = 2;
    if (v[o1][1] > v[o0][1]) {
        const int t = o0;
        o0 = o1;
        o1 = t;
    }
    if (v[o2][1] > v[o0][1]) {
        const int t = o2;
        o2 = o1;
        o1 = o0;
        o0 = t;
    } else if (v[o2][1] > v[o1][1]) {
        const int t = o1;
        o1 = o2;
        o2 = t;
    }
    const double* p0 = v[o0];
    const double* p1 = v[o1];
    const double* p2 = v[o2];

    // plane.cpp:66-89
    const double side = edge_side(p0, p2, p1);
    const bool up_left = (p0[0] >= p2[0]) && (side >= 0);
    const bool down_right = (p0[0] < p2[0]) && (side > 0);
    const bool long_edge_is_left = !(up_left || down_right);

    // plane.cpp:96-97 (double -> size_t conversions of non-negative values)
    const long long row_hi = static_cast<long long>(floor(frac_y(im, p0[1])));
    const long long row_lo = static_cast<long long>(ceil(frac_y(im, p2[1])));
    const uint32_t value = first_id + static_cast<uint32_t>(tet) + 1u;

    for (long long row = row_lo; row <= row_hi; ++row) {
        const double y = Ytab[row];  // == _lines[0][row_lo].y() + k * step_y accumulated (plane.cpp:100,138)
        const double* lower_a = (y < p1[1]) ? p2 : p0;
        const double x_long = edge_x_at(p0, p2, y);
        const double x_short = edge_x_at(lower_a, p1, y);
        const double x_lo = long_edge_is_left ? x_long : x_short;
        const double x_hi = long_edge_is_left ? x_short : x_long;
        const long long col_hi = static_cast<long long>(floor(frac_x(im, x_hi)));
        const long long col_lo = static_cast<long long>(ceil(frac_x(im, x_lo)));
        const int lrow = local_row_of(im, static_cast<int>(row));
        if (lrow < 0) continue;
        uint32_t* mrow = mask + static_cast<size_t>(lrow) * im.res_x;
        for (long long col = col_lo; col <= col_hi; ++col) atomicMax(mrow + col, value);
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

void launch_solid_mask_raster(hipStream_t s, const double* tets, int64_t n_tets, uint32_t first_id,
                              const double* Ytab, const ImageParams& im, uint32_t* mask) {
    if (n_tets <= 0) return;
    const unsigned blocks = static_cast<unsigned>((4 * n_tets + 255) / 256);
    hipLaunchKernelGGL(solid_mask_raster, dim3(blocks), dim3(256), 0, s, tets, n_tets, first_id, Ytab,
                       im, mask);
}

}  // namespace c5
