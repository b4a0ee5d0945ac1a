// Device helpers shared by the walk kernels (walk_kernels.hip: fp64, bit-faithful; walk_mixed.hip: fp32
// geometry with fp64 accumulators): face planes and per-cell records, tile shapes, exp, entry lists.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>

#include "device_types.hpp"
#include "kernels.hpp"

namespace c5 {

// ------------------------------------------------------------------------------------------
// build_records
// ------------------------------------------------------------------------------------------
struct FacePlane {
    double c, gx, gy;  // z = c + gx (x - x0) + gy (y - y0)
    int kind;          // +1 upper (cell body below the plane), -1 lower, 0 edge-on / flat
};

// Plane of face f of a cell with transformed vertices p[4], about the origin (x0, y0) = p[0].xy.
// Faces and vertex order as in the reference (plane.cpp:30-37, line.cpp:103-122):
//   0 = (0,1,2)  1 = (0,1,3)  2 = (0,2,3)  3 = (1,2,3);  z from line.cpp:158-171.
// Used by build_records and entry_raster so that an entry's z and the walk's z are the same numbers.
__device__ __forceinline__ FacePlane face_plane(const double (&p)[4][3], int f) {
    constexpr int FV[4][4] = {{0, 1, 2, 3}, {0, 1, 3, 2}, {0, 2, 3, 1}, {1, 2, 3, 0}};
    const double* a = p[FV[f][0]];
    const double* b = p[FV[f][1]];
    const double* c = p[FV[f][2]];
    const double* o = p[FV[f][3]];
    const double x0 = p[0][0], y0 = p[0][1];
    // line.cpp:158-171: z = ((y-ay)*A - (x-ax)*B)/m + az
    const double A = (b[0] - a[0]) * (c[2] - a[2]) - (c[0] - a[0]) * (b[2] - a[2]);
    const double B = (b[1] - a[1]) * (c[2] - a[2]) - (c[1] - a[1]) * (b[2] - a[2]);
    const double m = (b[0] - a[0]) * (c[1] - a[1]) - (c[0] - a[0]) * (b[1] - a[1]);
    FacePlane r;
    r.gy = A / m;
    r.gx = -B / m;
    r.c = a[2] + r.gx * (x0 - a[0]) + r.gy * (y0 - a[1]);
    const double z_under_opp = r.c + r.gx * (o[0] - x0) + r.gy * (o[1] - y0);
    const bool finite = (fabs(r.gx) <= DBL_MAX) && (fabs(r.gy) <= DBL_MAX) && (fabs(r.c) <= DBL_MAX);
    r.kind = (!finite || !(o[2] != z_under_opp)) ? 0 : (o[2] < z_under_opp ? 1 : -1);
    return r;
}

__device__ __forceinline__ bool build_cell_impl(const GridView& g, double alpha_limit, int order, int64_t cell,
                                                CellRecord& r, CellOptics& o, double (*verts)[3]);
// verts (optional): the cell's four transformed vertices
__device__ __forceinline__ bool build_cell(const GridView& g, double alpha_limit, int order, int64_t cell, CellRecord& r,
                                           CellOptics& o, double (*verts)[3] = nullptr) {
    return build_cell_impl(g, alpha_limit, order, cell, r, o, verts);
}

// Records of one cell; false if the cell is outside this context's row band (nothing to store).
__device__ __forceinline__ bool build_cell_impl(const GridView& g, double alpha_limit, int order, int64_t cell,
                                                CellRecord& r, CellOptics& o, double (*verts)[3]) {
    const int4 cv = g.cell_vert[cell];
    const int vid[4] = {cv.x, cv.y, cv.z, cv.w};
    double p[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k][1] = g.vy[vid[k]];
    // A ray of this context can only reach cells whose y-extent meets the band of its rows: skip the
    // rest (their records are never read).  This is what shards the per-view setup across ranks.
    const double cy_lo = fmin(fmin(p[0][1], p[1][1]), fmin(p[2][1], p[3][1]));
    const double cy_hi = fmax(fmax(p[0][1], p[1][1]), fmax(p[2][1], p[3][1]));
    if (cy_hi < g.cull_y_lo || cy_lo > g.cull_y_hi) return false;
    const int4 adj = g.cell_adj[cell];
    const int nb[4] = {adj.x, adj.y, adj.z, adj.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        p[k][0] = g.vx[vid[k]];
        p[k][2] = g.vz[vid[k]];
    }

    if (verts) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            verts[k][0] = p[k][0];
            verts[k][1] = p[k][1];
            verts[k][2] = p[k][2];
        }
    }
    FacePlane fp[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) fp[f] = face_plane(p, f);
    int n_true_up = 0, n_lo = 0, n_edge_on = 0;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        n_true_up += (fp[f].kind > 0) ? 1 : 0;
        n_lo += (fp[f].kind < 0) ? 1 : 0;
        n_edge_on += (fp[f].kind == 0) ? 1 : 0;
    }
    const int n_up = n_true_up + n_edge_on;  // edge-on faces ride along in the upper group (plane +inf)
    const bool flat = (n_lo == 0) || (n_true_up == 0);

    // walk order: upper (and edge-on) faces first, lower faces last; everything selected, nothing indexed
    r.x0 = p[0][0];
    r.y0 = p[0][1];
    int up_pos = 0, lo_pos = n_up;
    uint32_t words[4] = {kNoCell, kNoCell, kNoCell, kNoCell};
    double pl[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) pl[j][0] = pl[j][1] = pl[j][2] = 0.0;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const bool up = fp[f].kind >= 0;
        const int pos = up ? up_pos : lo_pos;
        up_pos += up ? 1 : 0;
        lo_pos += up ? 0 : 1;
        // an edge-on face is crossed by no ray: its slot forwards nothing (the LDS walk relies on this
        // to tell "left the grid" from the neighbour word alone)
        const uint32_t w = (nb[f] < 0 || fp[f].kind == 0) ? kNoCell : (static_cast<uint32_t>(nb[f]) & kIdMask);
        const double c = (fp[f].kind == 0) ? INFINITY : fp[f].c;
        const double gx = (fp[f].kind == 0) ? 0.0 : fp[f].gx;
        const double gy = (fp[f].kind == 0) ? 0.0 : fp[f].gy;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j == pos) {
                pl[j][0] = c;
                pl[j][1] = gx;
                pl[j][2] = gy;
                words[j] = w;
            }
        }
    }
    int stored_up = n_up;
    if (flat) {
        // no interior along z: neither contributes nor forwards the ray
        pl[0][0] = INFINITY;
        pl[0][1] = pl[0][2] = 0.0;
        pl[3][0] = -INFINITY;
        pl[3][1] = pl[3][2] = 0.0;
        pl[1][0] = INFINITY;
        pl[1][1] = pl[1][2] = 0.0;
        pl[2][0] = -INFINITY;
        pl[2][1] = pl[2][2] = 0.0;
        stored_up = 2;
        words[0] = words[1] = words[2] = words[3] = kNoCell;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r.plane[j][0] = pl[j][0];
        r.plane[j][1] = pl[j][1];
        r.plane[j][2] = pl[j][2];
        r.nbr[j] = words[j];
    }
    r.nbr[0] |= static_cast<uint32_t>(stored_up) << kUpperCountShift;

    // line.cpp:204-224
    const double a_raw = g.alpha[cell];
    const double qv = g.q[cell];
    double a_c = a_raw;
    if (a_c > alpha_limit) a_c = alpha_limit;
    o.alpha_raw = a_raw;
    o.q = qv;
    if (a_c < DBL_EPSILON) {
        o.alpha_c = 0.0;
        o.aux = 0.0;
    } else {
        o.alpha_c = a_c;
        o.aux = (order == 0) ? 1.0 / a_c : qv / a_c;
    }
    return true;
}

template <int TILE>
struct TileShape;
template <>
struct TileShape<0> {  // each wavefront owns a 64x1 row tile; workgroup 64 x 4
    static constexpr int WW = 64, WH = 1, GX = 1, GY = 4;
};
template <>
struct TileShape<1> {  // 16x4 per wavefront; workgroup 32 x 8
    static constexpr int WW = 16, WH = 4, GX = 2, GY = 2;
};
template <>
struct TileShape<2> {  // 8x8 per wavefront; workgroup 16 x 16
    static constexpr int WW = 8, WH = 8, GX = 2, GY = 2;
};

__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

struct alignas(16) D2 {
    double a, b;
};

// exp(x) for x <= 0 (x = -alpha * dz).  Same scheme as the device library's exp — k = rint(x log2 e),
// r = x - k ln 2 in two pieces, polynomial in r, scale by 2^k — but written as one fused
// multiply-add per coefficient with the coefficients as scalar constants: the library version keeps
// its eleven coefficients in vector registers (20 VGPRs of this kernel's 128) and pays a register
// copy per Horner step.  Taylor coefficients to r^13: truncation 4e-18 for |r| <= ln2 / 2, result
// within ~1 ulp.  Underflows to 0 below -745 like exp().
__device__ __forceinline__ double exp_nonpositive(double x) {
#pragma clang fp contract(fast)
    x = fmax(x, -746.0);  // k >= -1076: p * 2^k rounds to 0 there, as exp() does below -745.13
    const double k = rint(x * 1.4426950408889634074);  // log2(e)
    double r = fma(k, -6.93147180369123816490e-01, x);  // ln2 high part (exact product for |k| < 2^10)
    r = fma(k, -1.90821492927058770002e-10, r);         // ln2 low part
    // p = p * r + c with c in a scalar register pair: hipcc on its own keeps every coefficient in a
    // vector register pair and emits v_mov_b64 + v_fmac_f64 per step
    auto step = [](double acc, double rr, double c) {
        double out;
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(out) : "v"(acc), "v"(rr), "s"(c));
        return out;
    };
    double p = step(1.0 / 6227020800.0, r, 1.0 / 479001600.0);  // 1/13! r + 1/12!
    p = step(p, r, 1.0 / 39916800.0);
    p = step(p, r, 1.0 / 3628800.0);
    p = step(p, r, 1.0 / 362880.0);
    p = step(p, r, 1.0 / 40320.0);
    p = step(p, r, 1.0 / 5040.0);
    p = step(p, r, 1.0 / 720.0);
    p = step(p, r, 1.0 / 120.0);
    p = step(p, r, 1.0 / 24.0);
    p = step(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, static_cast<int>(k));
}

// exp(x) for -1/8 < x <= 0 without range reduction: Taylor to x^10 (truncation 2.9e-18), ten fused
// multiply-adds against the twenty instructions of exp_nonpositive.  A chord through one cell of a grid
// that resolves the image is short: alpha * dz stays below 0.08 on the whole C3 frame, so the walk takes this
// path whenever every lane of the wavefront qualifies (one ballot) and the general one otherwise.
__device__ __forceinline__ double exp_small_nonpositive(double r) {
    auto step = [](double acc, double rr, double c) {
        double out;
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(out) : "v"(acc), "v"(rr), "s"(c));
        return out;
    };
    double p = step(1.0 / 3628800.0, r, 1.0 / 362880.0);  // 1/10! r + 1/9!
    p = step(p, r, 1.0 / 40320.0);
    p = step(p, r, 1.0 / 5040.0);
    p = step(p, r, 1.0 / 720.0);
    p = step(p, r, 1.0 / 120.0);
    p = step(p, r, 1.0 / 24.0);
    p = step(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return p;
}
constexpr double kSmallExpArg = -0.125;

// touched once per frame: kept from displacing the cell records in L2 / Infinity Cache
__device__ __forceinline__ EntryHead load_entry_head(const EntryHead* p) {
    const long long v = __builtin_nontemporal_load(reinterpret_cast<const long long*>(p));
    EntryHead h;
    h.count = static_cast<int32_t>(v);
    h.chain = static_cast<int32_t>(v >> 32);
    return h;
}

// next place the ray enters the grid beyond s_cur (s = z walking down, -z walking up); -1 if none.
// The pixel's entries are first[lp] and the chain through the overflow pool (entry_raster).
template <bool kUp>
__device__ __forceinline__ int next_entry(const WalkParams& P, size_t lp, EntryHead h, double& s_cur) {
    double s_best = -DBL_MAX;
    int cell = -1;
    const Entry* e = P.entry_first + lp;
    int hop = h.chain;
    for (int k = 0; k < h.count; ++k) {  // bounded by the count: a chain cut short by a pool overflow ends at hop 0
        const double z = e->z;
        const int c = e->cell;
        const double se = kUp ? -z : z;
        if (se < s_cur && se > s_best) {
            s_best = se;
            cell = c;
        }
        if (hop <= 0 || hop > P.pool_capacity) break;
        e = P.entry_pool + (hop - 1);
        hop = e->next;
    }
    if (cell >= 0) s_cur = s_best;
    return cell;
}


}  // namespace c5
