// Device helpers of the walk kernels (walk_kernels.hip, fp64): face planes and per-cell records, tile shapes, exp,
// entry lists.
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

template <bool kOptics>
__device__ __forceinline__ bool build_cell_impl(const GridView& g, double alpha_limit, int order, int64_t cell,
                                                CellRecord& r, CellOptics& o, double (*verts)[3], int4* adj_out = nullptr);
// verts (optional): the cell's four transformed vertices
__device__ __forceinline__ bool build_cell(const GridView& g, double alpha_limit, int order, int64_t cell, CellRecord& r,
                                           CellOptics& o, double (*verts)[3] = nullptr) {
    return build_cell_impl<true>(g, alpha_limit, order, cell, r, o, verts);
}

// A cell's optics: its scalars, the alpha limit and the integration order — nothing of the view (line.cpp:204-224).
// The host has them rebuilt only when one of those changes (c_api.hip: optics_valid), not every frame.
__device__ __forceinline__ CellOptics cell_optics(const GridView& g, double alpha_limit, int order, int64_t cell) {
    CellOptics o;
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
    return o;
}

// Records of one cell; false if the cell is outside this context's row band (nothing to store).
template <bool kOptics>
__device__ __forceinline__ bool build_cell_impl(const GridView& g, double alpha_limit, int order, int64_t cell,
                                                CellRecord& r, CellOptics& o, double (*verts)[3], int4* adj_out) {
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
    if (adj_out) *adj_out = adj;
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

    if (kOptics) o = cell_optics(g, alpha_limit, order, cell);  // line.cpp:204-224
    return true;
}

// The cell's ExitRecord (device_types.hpp) from its walk-order record: the candidates a ray can LEAVE through — the upper
// slots walking along +z (order 0), the lower ones walking along -z (order 1, stored negated: w = -z) — as planes about
// the absolute pixel coordinates, and the optics.
__device__ __forceinline__ void to_exit_record(const CellRecord& r, const CellOptics& o, int order, ExitRecord& x) {
    const int n_up = static_cast<int>(r.nbr[0] >> kUpperCountShift);  // slots 0 .. n_up - 1 upper, the rest lower
    const double sign = order == 0 ? 1.0 : -1.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // order 0: candidates = slots 0, 1, 2 while they are upper; order 1: slots 3, 2, 1 while they are lower
        const int slot = order == 0 ? k : 3 - k;
        const bool is_candidate = order == 0 ? (slot < n_up) : (slot >= n_up);
        double c = INFINITY, gx = 0.0, gy = 0.0;
        uint32_t w = kNoCell;
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // selects, not runtime indexing
            if (j == slot && is_candidate) {
                const double pc = r.plane[j][0], pgx = r.plane[j][1], pgy = r.plane[j][2];
                if (fabs(pc) <= DBL_MAX) {  // (an edge-on slot or a flat cell carries +-inf: never an exit)
                    gx = sign * pgx;
                    gy = sign * pgy;
                    c = sign * (pc - pgx * r.x0 - pgy * r.y0);
                    w = r.nbr[j] & kIdMask;
                }
            }
        }
        x.plane[k][0] = c;
        x.plane[k][1] = gx;
        x.plane[k][2] = gy;
        x.nbr[k] = w;
    }
    x.flags = 0u;
    x.pad = 0.0;
    x.alpha_raw = o.alpha_raw;
    x.alpha_c = o.alpha_c;
    x.aux = o.aux;
    x.q = o.q;
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
template <>
struct TileShape<3> {  // 8x8 per wavefront, ONE wavefront per workgroup: a wavefront's slot (and its LDS) is free for
                       // the next tile the moment its own rays are done, not when the slowest of four is
    static constexpr int WW = 8, WH = 8, GX = 1, GY = 1;
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

// Next place the ray enters the grid beyond w_cur, in the walk coordinate w (w = z walking along +z, order 0; w = -z
// walking along -z; w grows along the walk): the cell, or -1 if there is none.  The pixel's entries are first[lp] and
// the chain through the overflow pool (entry_raster).
//
// Entries are compared by a depth KEY, not by the entry face's depth: the depth pushed a slack along the walk, INTO
// the cell the face belongs to (P.key_slack, the same for every face of the frame; times 2^k for a face steep
// against the rays, k in the entry's cell word).  Where two cells meet at faces that do not match (hanging nodes: a
// coarse face against several fine ones — conforming in space, not in connectivity; the reference never looks at
// connectivity, object3d_base.cpp:37-42 + plane.cpp:184-192 + line.cpp:138) both faces are boundary faces, the ray
// leaves through one and has to enter through the other AT THE SAME DEPTH — the same plane evaluated from two
// different vertex triples, equal only to rounding.  With the face's own depth as the key a strict comparison is a
// coin toss per crossing, and a lost toss loses the rest of the ray.  With the key pushed into the cell, "the nearest
// key beyond where the ray left the grid" finds the abutting cell whichever way the rounding fell, and never an entry
// already used: w_cur is moved to the key taken, and keys are compared strictly.
// On return w_entry = the entry's OWN depth (where the chord through its cell starts).
//
// Interpenetrating components (round 4).  Two parts of a grid that share no face but overlap in space — the reference bins
// and sorts any soup (plane.cpp:184-192, line.cpp:138) — cannot be walked: inside the overlap a ray is in two cells at
// once.  Connectivity does not show it, but the entry lists do: such a ray meets a boundary entry INSIDE a stretch it
// has just walked through cells — beyond the entry it took last (key_taken: that entry's key; -DBL_MAX for a ray that
// started from a cutting plane), before the depth at which it left the grid (stretch_hi) — an entry it can never take.
// `skipped` is set for it, with a margin of twice the entry's own key slack either side (keeps out the twin entries of
// a pixel centre exactly on the common edge of two boundary faces, and the abutting entry of a hanging-node interface,
// whose depth is that of the exit to rounding); the walk counts such rays, and the host answers with C5_RETRY and
// renders the grid with bin_sort_resolve from then on (c_api.hip: finish_frame).  Only on the (re-)entry path: nothing
// is added to a step.
//
// "depth_split": a job owns the entries whose own depth lies in [own_lo, own_hi) — below own_lo by no more than the
// entry's key slack: an abutting entry across a cutting plane may evaluate a rounding error below it — so that the
// jobs of a ray take every entry once (device_types.hpp: SplitParams; whole rays: -DBL_MAX, DBL_MAX); entries below
// own_lo belong to the stretch of a job further down and are not judged here.
template <bool kUp>
// has_pre: the pixel's first entry (pre_z, pre_cell) was loaded by the caller ahead of time (the walk's start: beside the
// entry head; by value - a pointer to a local would put it in scratch)
__device__ __forceinline__ int next_entry(const WalkParams& P, size_t lp, EntryHead h, double& w_cur, double& w_entry,
                                          double key_taken, double stretch_hi, bool& skipped, double own_lo = -DBL_MAX,
                                          double own_hi = DBL_MAX, bool has_pre = false, double pre_z = 0.0, uint32_t pre_cell = 0u) {
    double key_best = DBL_MAX;
    int cell = -1;
    const Entry* e = P.entry_first + lp;
    int hop = h.chain;
    for (int k = 0; k < h.count; ++k) {  // bounded by the count: a chain cut short by a pool overflow ends at hop 0
        const double z = (has_pre && k == 0) ? pre_z : e->z;
        const uint32_t word = (has_pre && k == 0) ? pre_cell : e->cell;
        const double we = kUp ? z : -z;
        const double slack = ldexp(P.key_slack, static_cast<int>(word >> kEntrySlackShift));
        const double key = we + slack;
        if (key > w_cur && key < key_best && we < own_hi && key >= own_lo) {
            key_best = key;
            cell = static_cast<int>(word & kIdMask);
            w_entry = we;
        }
        if (key > key_taken + 2.0 * slack && we + 2.0 * slack < stretch_hi && we > own_lo) skipped = true;
        if (hop <= 0 || hop > P.pool_capacity) break;
        e = P.entry_pool + (hop - 1);
        hop = e->next;
    }
    if (cell >= 0) w_cur = key_best;
    return cell;
}


// ------------------------------------------------------------------------------------------
// entry_raster: one wavefront per boundary face, ONE pass.
//
// Per covered pixel: old = atomicAdd(head[pixel].count, 1).  The first entry of a pixel (old == 0)
// goes straight into the dense array first[pixel] as one full 16-byte store; every further one
// (re-entry of a non-convex grid: rare) takes a slot of the overflow pool and hooks itself into the
// pixel's chain with one atomicExch on head[pixel].chain (slot + 1; 0 ends a chain).  The walk reads
// head[pixel], first[pixel] and the chain.  No scan over the pixels and no second raster pass
// (count -> scan -> fill took 35 + 30 + 52 us on the C3 frame); first[] is never cleared, only the
// 8-byte heads are.
// ------------------------------------------------------------------------------------------
// How far an entry's depth key lies behind its face (see next_entry): ONE slack for every face of the frame, times a
// power of two only for faces steep against the rays.
//   * uniform: keys shifted by the same amount keep the order of the true depths, so the walk takes a ray's entries in
//     exactly their order whatever the size of the slack — also a stretch of grid thinner than the slack that is
//     followed by another entry within it (a first version with a slack per face, proportional to the face's extent,
//     swapped two such entries in one fuzz scene in 400: a 4.6e-7 chord lost, S off by one).  It only has to exceed
//     the rounding between two evaluations of one plane from different vertex triples: `base` = 2^-24 of the grid's
//     bounding-box diagonal (1e-7 on the C3 grid, 1e5 times that rounding; c_api.hip).  Tied to the GRID's size: a
//     slack longer than a whole ray would let
//     a pixel that two boundary faces both claim (centre exactly on their common edge) walk the same cells twice.
//   * steep faces: a plane's depth is c + gx x + gy y with |gx x| of the size kappa * |x|: rounding eps * kappa * |x|,
//     and kappa * (the relative error of gx) * extent = eps * kappa^2 * extent on top, kappa^2 = 1 + gx^2 + gy^2
//     (times 256: the cell across the interface may be larger; capped at 2^-10 of the extent).  Only where THAT exceeds
//     the uniform slack does a face get more: the smallest k with base * 2^k above it (k <= 15) — those keys can swap
//     with a neighbour closer than their slack, a face a pixel hits with probability ~1 / kappa.
// Too small a slack loses the rest of a ray at a non-matching interface; too large a one costs nothing but the swap
// above.  base <= 0: no slack at all (option "entry_key" 0, testing: the behaviour before round 3).
__device__ __forceinline__ uint32_t entry_key_exponent(double gx, double gy, double extent, double coord, double base) {
    if (!(base > 0.0)) return 0u;
    const double kappa2 = fma(gx, gx, fma(gy, gy, 1.0));
    const double steep = fmin(256.0 * DBL_EPSILON * kappa2, 0x1p-10) * extent + 64.0 * DBL_EPSILON * sqrt(kappa2) * coord;
    if (!(steep > base)) return 0u;
    int e = 0;
    (void)frexp(steep / base, &e);  // steep / base = m * 2^e, 0.5 <= m < 1: 2^e >= the ratio
    return static_cast<uint32_t>(e < 1 ? 1 : (e > 15 ? 15 : e));
}

// entry_key_exponent of a face from its three vertices (extent: of the face along x, y and z; coord: its largest |x|, |y|)
__device__ __forceinline__ uint32_t face_key_exponent(double ax, double ay, double az, double bx, double by, double bz, double cx, double cy,
                                                      double cz, double gx, double gy, double base) {
    const double xmin = fmin(ax, fmin(bx, cx)), xmax = fmax(ax, fmax(bx, cx));
    const double ymin = fmin(ay, fmin(by, cy)), ymax = fmax(ay, fmax(by, cy));
    const double extent = (xmax - xmin) + (ymax - ymin) + (fmax(az, fmax(bz, cz)) - fmin(az, fmin(bz, cz)));
    const double coord = fmax(fmax(fabs(xmin), fabs(xmax)), fmax(fabs(ymin), fabs(ymax)));
    return entry_key_exponent(gx, gy, extent, coord, base);
}

struct RasterArgs {
    const double* Xtab;
    const double* Ytab;
    ImageParams im;
    EntryHead* head;
    Entry* first;
    Entry* pool;
    int64_t capacity;
    FrameCounters* counters;
    unsigned* sticky;
    int want_upper;
    double key_slack;  // the frame's uniform entry-key slack (entry_key_exponent's base; <= 0: none)
    // Which 8x8 pixel tiles (counted from the context's first row) hold an entry at all: a tile's word = tile_stamp.  The walk
    // asks for it first - one scalar load - and a wavefront whose tile has none stores its zeros without reading 64 entry
    // heads; one that has asks for heads, first entries and coordinates in ONE round of loads (walk_kernels.hip).  nullptr: not kept.
    uint32_t* tile_flag;
    uint32_t tile_stamp;
    int32_t tile_cols;  // tiles per row of tiles
};

__device__ __forceinline__ void raster_face(const RasterArgs& A, int lane, int64_t face_idx, double ax, double ay, double bx, double by,
                                            double cx, double cy, double x0, double y0, double pc, double pgx, double pgy, uint32_t cell_word);

// The work of one workgroup (four boundary faces); `block` is its index among the raster workgroups, so that
// the same body serves the stand-alone kernel and the fused per-view setup launch.  This form finds a face's vertices
// itself (boundary face -> cell -> four vertex ids -> twelve coordinates: three dependent rounds of loads);
// entry_raster_rec below starts from the record build_records left for the face.
__device__ __forceinline__ void entry_raster_block(const GridView& g, const RasterArgs& A, unsigned block) {
    const int want_upper = A.want_upper;
    const int lane = threadIdx.x & 63;
    const int64_t face_idx = block * 4ll + (threadIdx.x >> 6);
    if (face_idx >= g.n_bfaces) return;
    const uint32_t bf = g.bface[face_idx];
    const uint32_t cell = bf >> 2;
    const int f = static_cast<int>(bf & 3u);
    const int4 cv = g.cell_vert[cell];
    const int vid[4] = {cv.x, cv.y, cv.z, cv.w};
    double p[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k][1] = g.vy[vid[k]];
    // same band test as build_records: a culled cell has no record and no ray of this context
    if (fmax(fmax(p[0][1], p[1][1]), fmax(p[2][1], p[3][1])) < g.cull_y_lo ||
        fmin(fmin(p[0][1], p[1][1]), fmin(p[2][1], p[3][1])) > g.cull_y_hi)
        return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        p[k][0] = g.vx[vid[k]];
        p[k][2] = g.vz[vid[k]];
    }
    FacePlane fp;
    switch (f) {  // constant face index per case keeps the vertex selection in registers
        case 0: fp = face_plane(p, 0); break;
        case 1: fp = face_plane(p, 1); break;
        case 2: fp = face_plane(p, 2); break;
        default: fp = face_plane(p, 3); break;
    }
    // walking from +z to -z a ray enters through faces the cell body lies below (upper faces);
    // walking from -z to +z through the others
    if (fp.kind == 0 || ((fp.kind > 0) != (want_upper != 0))) return;

    const int i0 = (f == 3) ? 1 : 0;
    const int i1 = (f <= 1) ? 1 : 2;
    const int i2 = (f == 0) ? 2 : 3;
    double ax, ay, az, bx, by, bz, cx, cy, cz;
    {   // selects, not runtime indexing
        ax = i0 == 0 ? p[0][0] : p[1][0];
        ay = i0 == 0 ? p[0][1] : p[1][1];
        az = i0 == 0 ? p[0][2] : p[1][2];
        bx = i1 == 1 ? p[1][0] : p[2][0];
        by = i1 == 1 ? p[1][1] : p[2][1];
        bz = i1 == 1 ? p[1][2] : p[2][2];
        cx = i2 == 2 ? p[2][0] : p[3][0];
        cy = i2 == 2 ? p[2][1] : p[3][1];
        cz = i2 == 2 ? p[2][2] : p[3][2];
    }

    // an entry carries its face's own depth; how far behind it the entry is KEYED is the frame's uniform slack times
    // 2^k, k > 0 only for a face steep against the rays: once per face, in the top bits of the cell word
    const uint32_t cell_word = cell | (face_key_exponent(ax, ay, az, bx, by, bz, cx, cy, cz, fp.gx, fp.gy, A.key_slack) << kEntrySlackShift);
    raster_face(A, lane, face_idx, ax, ay, bx, by, cx, cy, p[0][0], p[0][1], fp.c, fp.gx, fp.gy, cell_word);
}

// The pixels of one boundary face (one wavefront): see entry_raster above.  (x0, y0): the origin the face's plane
// (pc, pgx, pgy) is written about.
__device__ __forceinline__ void raster_face(const RasterArgs& A, int lane, int64_t face_idx, double ax, double ay, double bx, double by,
                                            double cx, double cy, double x0, double y0, double pc, double pgx, double pgy, uint32_t cell_word) {
    const double* __restrict__ Xtab = A.Xtab;
    const double* __restrict__ Ytab = A.Ytab;
    const ImageParams& im = A.im;
    EntryHead* __restrict__ head = A.head;
    Entry* __restrict__ first = A.first;
    Entry* __restrict__ pool = A.pool;
    const int64_t capacity = A.capacity;
    FrameCounters* counters = A.counters;
    unsigned* sticky = A.sticky;
    const double xmin = fmin(ax, fmin(bx, cx)), xmax = fmax(ax, fmax(bx, cx));
    const double ymin = fmin(ay, fmin(by, cy)), ymax = fmax(ay, fmax(by, cy));
    // conservative pixel box: floor / ceil already include a pixel on either side whose centre lies outside
    // the face (the pixel coordinates are running sums, off from x_min + i * step by ~1e-13 of a pixel: far
    // less than that margin needs)
    double fc0 = floor((xmin - im.x_min) / im.step_x);
    double fc1 = ceil((xmax - im.x_min) / im.step_x);
    double fr0 = floor((ymin - im.y_min) / im.step_y);
    double fr1 = ceil((ymax - im.y_min) / im.step_y);
    if (!(fc1 >= 0.0) || !(fr1 >= 0.0) || !(fc0 <= im.res_x - 1.0) || !(fr0 <= im.res_y - 1.0)) return;
    const int c0 = static_cast<int>(fmax(fc0, 0.0));
    const int c1 = static_cast<int>(fmin(fc1, im.res_x - 1.0));
    const int r0 = max(static_cast<int>(fmax(fr0, 0.0)), im.row_begin);
    const int r1 = min(static_cast<int>(fmin(fr1, im.res_y - 1.0)), im.row_begin + im.row_count - 1);
    if (r1 < r0) return;
    // only the rows of this context (cyclic row tiles: every world-th tile), as a range of LOCAL rows
    int lr0, lr1;
    local_row_span(im, r0, r1, lr0, lr1);
    if (lr1 < lr0) return;
    const unsigned bw = static_cast<unsigned>(c1 - c0 + 1);
    const unsigned n_box = bw * static_cast<unsigned>(lr1 - lr0 + 1);  // <= pixels of the image: fits 32 bits
    if (A.tile_flag) {
        // every 8x8 tile the face's BOX meets is marked (a superset of the tiles that get an entry: a marked tile without one
        // is read and found empty, an unmarked one is never read) - one store per face and lane, not one per pixel
        const int tx0 = c0 >> 3, ty0 = lr0 >> 3;
        const unsigned ntx = static_cast<unsigned>((c1 >> 3) - tx0 + 1), n_t = ntx * static_cast<unsigned>((lr1 >> 3) - ty0 + 1);
        for (unsigned t = static_cast<unsigned>(lane); t < n_t; t += 64u) {
            const unsigned r = t / ntx;
            A.tile_flag[static_cast<size_t>(ty0 + static_cast<int>(r)) * A.tile_cols + tx0 + static_cast<int>(t - r * ntx)] = A.tile_stamp;
        }
    }

    // The raster is bound by vector instructions (the box of a face holds 2.5x the pixels of the face), so
    // the per-pixel work is kept small: the three edge functions as planes about vertex a (two fused
    // multiply-adds each; which of two faces claims a pixel within rounding of their common edge was never
    // defined by the older product form either) and the row / column split by a float reciprocal with a
    // one-step correction instead of an integer division.
    const double ea0 = -(by - ay), eb0 = bx - ax;                       // edge a -> b
    const double ea1 = -(cy - by), eb1 = cx - bx;                       // edge b -> c
    const double ec1 = -(ea1 * (bx - ax) + eb1 * (by - ay));
    const double ea2 = -(ay - cy), eb2 = ax - cx;                       // edge c -> a
    const double ec2 = -(ea2 * (cx - ax) + eb2 * (cy - ay));
    const float inv_bw = 1.0f / static_cast<float>(bw);
    const bool small_box = n_box < (1u << 24);  // float(idx) exact: the estimate is off by one at most
    // Two 64-pixel chunks of the box per iteration, their returning atomics in flight together.  Measured on the C3
    // frame / the C2 ball (entry_raster, us): 1 chunk 50 / 20.8, 2 chunks 47 / 20.6, 3 -> 49 / 21.3, 4 -> 52 / 22.3,
    // 8 -> 55 / 24.3: a face's box holds ~400 pixels, and wider iterations end in mostly idle ones.
#ifndef C5_RASTER_CHUNKS
#define C5_RASTER_CHUNKS 2
#endif
    constexpr int kChunks = C5_RASTER_CHUNKS;
    // The box's pixel coordinates once, one column and one row per lane, handed round the wavefront by lane permutes: read
    // from the tables per chunk they were a dependent load in front of every chunk's atomics (round 4; a box wider or
    // taller than 64 pixels - a face that spans a sixth of the image - reads the tables as before).
    const unsigned n_rows_box = static_cast<unsigned>(lr1 - lr0 + 1);
    const bool by_permute = bw <= 64u && n_rows_box <= 64u;
    double x_of_lane = 0.0, y_of_lane = 0.0;
    if (by_permute) {
        x_of_lane = Xtab[c0 + static_cast<int>(min(static_cast<unsigned>(lane), bw - 1u))];
        y_of_lane = Ytab[global_row_of(im, lr0 + static_cast<int>(min(static_cast<unsigned>(lane), n_rows_box - 1u)))];
    }
    for (unsigned base = 0; base < n_box; base += 64u * kChunks) {
        bool in[kChunks];
        int slot[kChunks];  // of the depth sample (DepthSamples), or -1
        size_t lp[kChunks];
        double z[kChunks];
        int old[kChunks];
#pragma unroll
        for (int k = 0; k < kChunks; ++k) {
            const unsigned idx = base + 64u * k + static_cast<unsigned>(lane);
            in[k] = false;
            slot[k] = -1;
            lp[k] = 0;
            z[k] = 0.0;
            if (base + 64u * k >= n_box) continue;  // (wave-uniform: the permutes below want every lane)
            {
                unsigned qrow, rcol;
                if (small_box) {
                    qrow = static_cast<unsigned>(static_cast<float>(idx) * inv_bw);
                    int rem = static_cast<int>(idx - qrow * bw);
                    if (rem < 0) {
                        qrow -= 1u;
                        rem += static_cast<int>(bw);
                    } else if (rem >= static_cast<int>(bw)) {
                        qrow += 1u;
                        rem -= static_cast<int>(bw);
                    }
                    rcol = static_cast<unsigned>(rem);
                } else {
                    qrow = idx / bw;
                    rcol = idx - qrow * bw;
                }
                const bool live = idx < n_box;
                if (!live) qrow = 0u, rcol = 0u;
                const int lrow = lr0 + static_cast<int>(qrow);
                const int col = c0 + static_cast<int>(rcol);
                double x, y;
                if (by_permute) {
                    x = __shfl(x_of_lane, static_cast<int>(rcol));
                    y = __shfl(y_of_lane, static_cast<int>(qrow));
                } else {
                    x = Xtab[col];
                    y = Ytab[global_row_of(im, lrow)];
                }
                if (live) {
                    // closed point-in-triangle test, either winding
                    const double dxa = x - ax, dya = y - ay;
                    const double e0 = fma(ea0, dxa, eb0 * dya);
                    const double e1 = fma(ea1, dxa, fma(eb1, dya, ec1));
                    const double e2 = fma(ea2, dxa, fma(eb2, dya, ec2));
                    in[k] = (e0 >= 0 && e1 >= 0 && e2 >= 0) || (e0 <= 0 && e1 <= 0 && e2 <= 0);
                    lp[k] = static_cast<size_t>(lrow) * im.res_x + col;
                    z[k] = pc + pgx * (x - x0) + pgy * (y - y0);
                    slot[k] = in[k] ? fit_slot_of(im, col, global_row_of(im, lrow)) : -1;

                }
            }
        }
#pragma unroll
        for (int k = 0; k < kChunks; ++k) old[k] = in[k] ? atomicAdd(&head[lp[k]].count, 1) : 0;
#pragma unroll
        for (int k = 0; k < kChunks; ++k) {
            if (in[k] && old[k] == 0) {
                Entry e;
                e.z = z[k];
                e.cell = cell_word;
                e.next = 0;
                first[lp[k]] = e;
                if (slot[k] >= 0)  // (in the walk coordinate)
                    reinterpret_cast<DepthSamples*>(counters + kCounterShards)->entry_key[slot[k]] = depth_key(A.want_upper ? -z[k] : z[k]);
            }
            // further entries: one pool allocation per wavefront and chunk (same-address atomics serialise)
            const bool more = in[k] && old[k] > 0;
            const unsigned long long more_mask = __builtin_amdgcn_ballot_w64(more);
            if (more_mask == 0ull) continue;
            // The pool is cut into kCounterShards parts, each with its counter on a line of its own (a
            // non-convex grid makes tens of thousands of these allocations per frame; on ONE word they
            // would serialise at ~10 ns each).  A request starts at its home shard and takes what that shard
            // has left; what it still lacks it asks of the next shard, and so on round the ring: a slot is
            // refused only when every shard is exhausted, i.e. a frame overflows if and only if its TOTAL
            // demand (sum over the pixels of entries - 1, a property of grid, view and image alone)
            // exceeds the capacity — never because of which faces hash to which shard or of the order
            // the entries of a pixel arrive in.  (Round 1 gave every shard a fixed capacity / 64: one
            // large re-entry face could overflow its shard with 63 others empty, and which entry of a
            // pixel needs a slot at all is a race, so the same frame overflowed or not from run to run.)
            const unsigned n_more = static_cast<unsigned>(__popcll(more_mask));
            const unsigned my_rank = static_cast<unsigned>(__popcll(more_mask & ((1ull << lane) - 1ull)));
            const int leader = __builtin_ctzll(more_mask);
            unsigned shard = static_cast<unsigned>(face_idx) % kCounterShards;
            unsigned taken = 0;  // wave-uniform: requests served so far
            long long slot = -1;
            for (int t = 0; t < kCounterShards && taken < n_more; ++t, shard = (shard + 1u) % kCounterShards) {
                const unsigned lo = static_cast<unsigned>((static_cast<unsigned long long>(shard) * static_cast<unsigned long long>(capacity)) / kCounterShards);
                const unsigned hi = static_cast<unsigned>((static_cast<unsigned long long>(shard + 1u) * static_cast<unsigned long long>(capacity)) / kCounterShards);
                const unsigned cap_s = hi - lo;
                const unsigned want = n_more - taken;
                unsigned base = cap_s;
                if (lane == leader) {
                    // The home shard is simply asked (one returning atomic, as before the ring).  On the way round
                    // the ring a shard known to be full is passed by without touching its counter (the counters
                    // only grow within a frame, so a stale reading errs on the side of asking).
                    unsigned* const used = &counters[shard].pool_used;
                    if (t == 0 || __hip_atomic_load(used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < cap_s) base = atomicAdd(used, want);
                }
                base = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(base), leader));
                const unsigned avail = base < cap_s ? min(want, cap_s - base) : 0u;
                if (more && my_rank >= taken && my_rank < taken + avail) slot = static_cast<long long>(lo) + base + (my_rank - taken);
                taken += avail;
            }
            if (more && slot >= 0) {
                Entry e;
                e.z = z[k];
                e.cell = cell_word;
                e.next = atomicExch(&head[lp[k]].chain, static_cast<int32_t>(slot) + 1);
                pool[slot] = e;
            }
            if (taken < n_more && lane == leader) {
                // pool exhausted: the frame is incomplete.  The host reads the number of entries that found
                // no slot (this frame: counters[0].entry_overflow; any frame since it last looked: sticky[0]),
                // grows the pool by at least that and reports C5_RETRY; the walk bounds-checks every hop,
                // so such a frame is wrong but never unsafe.
                atomicAdd(&counters->entry_overflow, n_more - taken);
                atomicAdd(sticky, n_more - taken);
            }
        }
    }
}


}  // namespace c5
