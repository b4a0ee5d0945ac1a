// gfx950 kernels of the render hot path (fp64; FMA contraction allowed here).
//
//   build_records    per view: cell -> ONE 128-byte ExitRecord (device_types.hpp): the (up to three) planes a ray
//                    can LEAVE the cell through, in the walk coordinate about the absolute pixel coordinates, the
//                    neighbour words behind them, and the cell's optics (alpha, clamped alpha, 1 / alpha or Q / alpha,
//                    Q).  Replaces the per-segment plane solve line::find_polygon_intersection_z
//                    (line.cpp:150-174) and the per-step clamp/divide of
//                    line::integrate_ray_value_by_i (line.cpp:213-224).
//   entry_raster     boundary faces facing the viewer -> per-pixel entry records (CSR).
//                    Replaces the part of plane::find_intersections (plane.cpp:184-192) that
//                    discovers where a ray meets the grid; re-entries of non-convex grids
//                    are further entries of the same pixel.
//   walk_composite   one lane per pixel: face-adjacency walk along z; tau and the
//                    emission/absorption integral, either back to front in the reference's own
//                    arithmetic (default) or front to back with a transmittance early-out.
//                    Replaces plane::trace_rays' loop body (plane.cpp:161-169):
//                    line::calculate_intersections + std::sort (line.cpp:84-148),
//                    direct_calculate_ray_value (line.cpp:176-193) and
//                    integrate_ray_value_by_i (line.cpp:195-227), fp32 store (plane.cpp:165-166).
#include <hip/hip_runtime.h>

#include <cfloat>

#include "device_types.hpp"
#include "kernels.hpp"
#include "walk_common.hpp"

namespace c5 {

// One thread builds one cell's record in registers; a wavefront then writes its 64 records through a
// wave-private LDS area so that every store instruction covers 1 KiB of consecutive addresses (a
// thread storing its own 128-byte record would touch 64 different lines per instruction).
struct alignas(16) Q4 {
    uint32_t a, b, c, d;
};
constexpr int kRecPad = 9;  // 16-byte units per record in LDS: 8 + 1 pad (conflict-free b128 rows)

// How far beyond a cutting plane a cell may begin and still be listed for it / claim a pixel of it (plane_raster): the
// rounding of a point-in-cell test, generously — a claim this far off moves the start of one chord by as much.
__device__ __forceinline__ double plane_tolerance(double coord, double extent) { return 64.0 * DBL_EPSILON * (coord + extent); }

template <bool SPLIT, bool BF>
__device__ __forceinline__ void build_records_block(const GridView& g, double alpha_limit, int order, unsigned block,
                                                    Q4 (*s_rec)[64 * kRecPad]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (g.block_sphere) {  // (uniform) all of this workgroup's cells outside the context's rows?  (kernels.hpp)
        const double4 s = g.block_sphere[block];
        double cx = s.x, cy = s.y, cz = s.z;
        for (int r = 0; r < g.rot.n; ++r) {  // tetra.cpp:44-62 on the centre (its rounding is covered by the margin below)
            const double co = g.rot.cosv[r], si = g.rot.sinv[r];
            if (g.rot.axis[r] == 0) {
                const double y_old = cy;
                cy = cy * co - cz * si;
                cz = y_old * si + cz * co;
            } else {
                cx -= g.rot.x0[r];
                const double x_old = cx;
                cx = cx * co - cz * si;
                cz = x_old * si + cz * co;
                cx += g.rot.x0[r];
            }
        }
        const double reach = s.w + 1e-9 * (s.w + fabs(cy) + 1.0);
        if (cy + reach < g.cull_y_lo || cy - reach > g.cull_y_hi) return;
    }
    const int64_t cell = block * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    const int64_t wave_first = cell - lane;
    bool valid = cell < g.n_cells;
    CellRecord r;
    CellOptics o;
    double verts[4][3];
    int4 adj = make_int4(0, 0, 0, 0);
    if (valid) valid = build_cell_impl<true>(g, alpha_limit, order, cell, r, o, (SPLIT || BF) ? verts : nullptr, BF ? &adj : nullptr);
    const unsigned long long valid_mask = __builtin_amdgcn_ballot_w64(valid);
    if (valid_mask == 0ull) return;  // wave-uniform
    if (BF) {
        // The cell's boundary faces a ray can ENTER through leave a record for entry_raster_rec (device_types.hpp:
        // BFaceRecord): the vertices and the face planes are in registers here, the raster would have to find them again
        // through three dependent rounds of loads (face -> cell -> vertex ids -> coordinates).  One wavefront in sixteen
        // has a boundary cell on the C3 grid.
        const bool b_cell = valid && (adj.x <= -2 || adj.y <= -2 || adj.z <= -2 || adj.w <= -2);
        if (__builtin_amdgcn_ballot_w64(b_cell) != 0ull && b_cell) {
            const int nbv[4] = {adj.x, adj.y, adj.z, adj.w};
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                if (nbv[f] > -2) continue;
                const FacePlane fp = face_plane(verts, f);
                // walking from +z to -z a ray enters through faces the cell body lies below (upper faces); walking from -z
                // to +z through the others; an edge-on face is entered by no ray
                if (fp.kind == 0 || ((fp.kind > 0) != (g.bf_want_upper != 0))) continue;
                constexpr int FV3[4][3] = {{0, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 2, 3}};
                const double* a = verts[FV3[f][0]];
                const double* b = verts[FV3[f][1]];
                const double* c = verts[FV3[f][2]];
                BFaceRecord rec;
                rec.ax = a[0], rec.ay = a[1], rec.bx = b[0], rec.by = b[1], rec.cx = c[0], rec.cy = c[1];
                rec.x0 = verts[0][0], rec.y0 = verts[0][1];
                rec.pc = fp.c, rec.pgx = fp.gx, rec.pgy = fp.gy;
                rec.cell_word = static_cast<uint32_t>(cell) |
                                (face_key_exponent(a[0], a[1], a[2], b[0], b[1], b[2], c[0], c[1], c[2], fp.gx, fp.gy, g.bf_key_slack) << kEntrySlackShift);
                rec.seq = g.bf_seq;
                g.bfrec[-nbv[f] - 2] = rec;
            }
        }
    }
    if (SPLIT) {
        // "depth_split": the cells that straddle a cutting plane go on the list plane_raster works through (one
        // allocation per wavefront and plane; the list has room for every cell at every plane)
        // (a vertex's depth against the planes' common tilt: plane pl passes the cell if it lies between the least and the
        // greatest of the four)
        double z_lo = 0.0, z_hi = 0.0, tol = 0.0;
        if (valid) {
            double d[4], coord = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = verts[k][2] - fma(g.split.gx, verts[k][0], g.split.gy * verts[k][1]);
                coord = fmax(coord, fmax(fabs(verts[k][0]), fmax(fabs(verts[k][1]), fabs(verts[k][2]))));
            }
            z_lo = fmin(fmin(d[0], d[1]), fmin(d[2], d[3]));
            z_hi = fmax(fmax(d[0], d[1]), fmax(d[2], d[3]));
            tol = plane_tolerance(coord * (1.0 + fabs(g.split.gx) + fabs(g.split.gy)), z_hi - z_lo);
        }
        // The list is cut into kStraddleShards parts, each with its counter on a line of its own (thousands of wavefronts
        // adding to ONE word would serialise at ~10 ns each); a wavefront belongs to the part (its index mod 64), and a
        // part has room for everything its wavefronts could ever list (64 cells x (K - 1) planes each): no overflow.
        const unsigned shard = static_cast<unsigned>((wave_first >> 6) % kStraddleShards);
        for (int pl = 1; pl < g.split.n_slabs; ++pl) {
            const double w = g.split.w[pl];
            const bool cut = valid && z_lo <= w + tol && z_hi >= w - tol;
            const unsigned long long cut_mask = __builtin_amdgcn_ballot_w64(cut);
            if (cut_mask == 0ull) continue;
            const int leader = __builtin_ctzll(cut_mask);
            unsigned base = 0;
            if (lane == leader) base = atomicAdd(g.split.straddle_count + shard * kStraddleCounterStride, static_cast<unsigned>(__popcll(cut_mask)));
            base = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(base), leader));
            const unsigned at = base + static_cast<unsigned>(__popcll(cut_mask & ((1ull << lane) - 1ull)));
            if (cut && at < g.split.straddle_capacity)
                g.split.straddle[static_cast<size_t>(shard) * g.split.straddle_capacity + at] = static_cast<uint32_t>(cell) | (static_cast<uint32_t>(pl - 1) << 28);
        }
    }
    Q4* const my_rec = s_rec[wave];
    if (valid) {
        ExitRecord x;
        to_exit_record(r, o, order, x);
        const Q4* rp = reinterpret_cast<const Q4*>(&x);
#pragma unroll
        for (int k = 0; k < 8; ++k) my_rec[lane * kRecPad + k] = rp[k];
    }
    __builtin_amdgcn_wave_barrier();
    Q4* const rec_out = reinterpret_cast<Q4*>(g.xrec + wave_first);
#pragma unroll
    for (int k = 0; k < 8; ++k) {  // 8 x 1 KiB: records 8k .. 8k+7 of the wavefront
        const int rec_i = k * 8 + (lane >> 3);
        if ((valid_mask >> rec_i) & 1ull) rec_out[k * 64 + lane] = my_rec[rec_i * kRecPad + (lane & 7)];
    }
}

template <bool SPLIT, bool BF>
__global__ __launch_bounds__(256) void build_records(GridView g, double alpha_limit, int order) {
    __shared__ Q4 s_rec[4][64 * kRecPad];
    build_records_block<SPLIT, BF>(g, alpha_limit, order, blockIdx.x, s_rec);
}

// entry_raster from the face records of build_records: one wavefront per boundary face, one scalar load of 96 bytes
// instead of three dependent rounds of gathers; a face without a record of THIS frame (turned away from the rays,
// edge-on, its cell outside this context's rows) costs that one load.
__global__ __launch_bounds__(256) void entry_raster_rec(GridView g, RasterArgs A) {
    const int lane = threadIdx.x & 63;
    const int64_t face_idx = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x * 4u + (threadIdx.x >> 6)));
    if (face_idx >= g.n_bfaces) return;
    const BFaceRecord* __restrict__ rp = g.bfrec + face_idx;
    if (rp->seq != g.bf_seq) return;
    raster_face(A, lane, face_idx, rp->ax, rp->ay, rp->bx, rp->by, rp->cx, rp->cy, rp->x0, rp->y0, rp->pc, rp->pgx, rp->pgy, rp->cell_word);
}

__global__ __launch_bounds__(256) void entry_raster(GridView g, RasterArgs A) { entry_raster_block(g, A, blockIdx.x); }

// The per-view setup as ONE launch: record workgroups (HBM-bound: 304 bytes per cell) and raster workgroups
// (bound by vector instructions and returning atomics) interleaved, so that the two kinds of work share the GPU
// instead of running one after the other.  Neither reads what the other writes (both read the transformed vertices).
__global__ __launch_bounds__(256) void setup_fused(GridView g, double alpha_limit, int order, unsigned n_rec, unsigned n_ras,
                                                   RasterArgs A) {
    __shared__ Q4 s_rec[4][64 * kRecPad];
    const unsigned b = blockIdx.x, m = n_rec < n_ras ? n_rec : n_ras;
    const bool raster = b < 2u * m ? (b & 1u) != 0u : n_ras > n_rec;
    const unsigned idx = b < 2u * m ? b >> 1 : b - m;
    if (raster)
        entry_raster_block(g, A, idx);
    else
        build_records_block<false, false>(g, alpha_limit, order, idx, s_rec);
}

// ------------------------------------------------------------------------------------------
// plane_raster ("depth_split"): where is every ray at the cutting planes?
//
// build_records has listed the cells that straddle a plane w = w[pl] (to a tolerance).  A wavefront takes one listed
// cell: the plane cuts it in a triangle or a quadrilateral; the pixels of the cell's bounding box are tested against
// the cell's four half-spaces at depth w, in NORMAL form — s_f(x, y) = n_f . (x, y, w) - d_f, oriented so that the
// cell's fourth vertex is on the positive side: no division, vertical faces are faces like any other — and a pixel
// inside gets plane_cell[pl - 1][pixel] = stamp << 28 | cell: the cell in which the job of slab pl starts that ray, at
// depth w (walk_composite_lds<..., SPLIT>).  Two claims:
//   * inside, every s_f >= 0: a plain store — cells that share the point (it lies on a common face) both contain it,
//     whichever store lands last is right;
//   * within the rounding of the test, every s_f >= -tol_f: taken only if nobody else has claimed the pixel (compare
//     and swap against a word of another frame) — so that a point within rounding of a face never falls between the
//     two cells that share it; such a start is off by the rounding of one plane evaluation.
// A pixel no cell claims is outside the grid at that depth: its word keeps an old stamp, and the job looks for the
// ray's next boundary entry instead.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void plane_raster(GridView g, const double* __restrict__ Xtab, const double* __restrict__ Ytab,
                                                    ImageParams im) {
    const int lane = threadIdx.x & 63;
    const unsigned gw = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = gridDim.x * 4u;
    // the shards' fill counts -> one running sum across the lanes (lane k: items of shards 0 .. k)
    unsigned incl = __hip_atomic_load(g.split.straddle_count + lane * kStraddleCounterStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    incl = incl < g.split.straddle_capacity ? incl : g.split.straddle_capacity;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned up = static_cast<unsigned>(__shfl_up(static_cast<int>(incl), d));
        if (lane >= d) incl += up;
    }
    const unsigned total = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
    if (gw == 0) g.split.straddle_count_next[lane * kStraddleCounterStride] = 0u;  // (the half the NEXT frame's build_records fills)
    for (unsigned item = gw; item < total; item += n_waves) {
        const unsigned shard = static_cast<unsigned>(__popcll(__builtin_amdgcn_ballot_w64(incl <= item)));
        const unsigned before = shard ? static_cast<unsigned>(__shfl(static_cast<int>(incl), static_cast<int>(shard) - 1)) : 0u;
        const uint32_t word = g.split.straddle[static_cast<size_t>(shard) * g.split.straddle_capacity + (item - before)];
        const uint32_t cell = word & kIdMask;
        const int pl = static_cast<int>(word >> 28) + 1;
        const double w = g.split.w[pl];
        const int4 cv = g.cell_vert[cell];
        const int vid[4] = {cv.x, cv.y, cv.z, cv.w};
        double p[4][3];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            p[k][0] = g.vx[vid[k]];
            p[k][1] = g.vy[vid[k]];
            p[k][2] = g.vz[vid[k]];
        }
        // the four half-spaces at depth w: s_f(x, y) = fa x + fb y + fc >= 0 inside
        constexpr int FV[4][4] = {{0, 1, 2, 3}, {0, 1, 3, 2}, {0, 2, 3, 1}, {1, 2, 3, 0}};
        double fa[4], fb[4], fc[4], ft[4];
        bool flat = false;
        double coord = fabs(w);
#pragma unroll
        for (int k = 0; k < 4; ++k) coord = fmax(coord, fmax(fabs(p[k][0]), fabs(p[k][1])));
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const double* a = p[FV[f][0]];
            const double* b = p[FV[f][1]];
            const double* c = p[FV[f][2]];
            const double* o = p[FV[f][3]];
            const double ux = b[0] - a[0], uy = b[1] - a[1], uz = b[2] - a[2];
            const double vx = c[0] - a[0], vy = c[1] - a[1], vz = c[2] - a[2];
            double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
            const double side = nx * (o[0] - a[0]) + ny * (o[1] - a[1]) + nz * (o[2] - a[2]);
            if (!(side != 0.0)) flat = true;  // (no volume, or NaN: claims nothing)
            if (side < 0.0) nx = -nx, ny = -ny, nz = -nz;
            // the point of pixel (x, y) on plane pl is (x, y, w + gx x + gy y): s_f stays linear in x and y
            fa[f] = fma(nz, g.split.gx, nx);
            fb[f] = fma(nz, g.split.gy, ny);
            fc[f] = nz * (w - a[2]) - nx * a[0] - ny * a[1];
            // rounding of s_f at a pixel: a few ulps of the largest term
            ft[f] = 32.0 * DBL_EPSILON * ((fabs(nx) + fabs(ny) + fabs(nz) * (1.0 + fabs(g.split.gx) + fabs(g.split.gy))) * coord);
        }
        if (flat) continue;
        const double xmin = fmin(fmin(p[0][0], p[1][0]), fmin(p[2][0], p[3][0])), xmax = fmax(fmax(p[0][0], p[1][0]), fmax(p[2][0], p[3][0]));
        const double ymin = fmin(fmin(p[0][1], p[1][1]), fmin(p[2][1], p[3][1])), ymax = fmax(fmax(p[0][1], p[1][1]), fmax(p[2][1], p[3][1]));
        // conservative pixel box, as in entry_raster
        const double fc0 = floor((xmin - im.x_min) / im.step_x), fc1 = ceil((xmax - im.x_min) / im.step_x);
        const double fr0 = floor((ymin - im.y_min) / im.step_y), fr1 = ceil((ymax - im.y_min) / im.step_y);
        if (!(fc1 >= 0.0) || !(fr1 >= 0.0) || !(fc0 <= im.res_x - 1.0) || !(fr0 <= im.res_y - 1.0)) continue;
        const int c0 = static_cast<int>(fmax(fc0, 0.0));
        const int c1 = static_cast<int>(fmin(fc1, im.res_x - 1.0));
        const int r0 = max(static_cast<int>(fmax(fr0, 0.0)), im.row_begin);
        const int r1 = min(static_cast<int>(fmin(fr1, im.res_y - 1.0)), im.row_begin + im.row_count - 1);
        if (r1 < r0) continue;
        int lr0, lr1;
        local_row_span(im, r0, r1, lr0, lr1);
        if (lr1 < lr0) continue;
        const unsigned bw = static_cast<unsigned>(c1 - c0 + 1);
        const unsigned n_box = bw * static_cast<unsigned>(lr1 - lr0 + 1);
        uint32_t* const words = g.split.plane_cell + static_cast<size_t>(pl - 1) * g.split.plane_stride;
        const uint32_t mine = (g.split.stamp << 28) | cell;
        for (unsigned base = 0; base < n_box; base += 64u) {
            const unsigned idx = base + static_cast<unsigned>(lane);
            if (idx >= n_box) continue;
            const unsigned qrow = idx / bw, rcol = idx - qrow * bw;
            const int lrow = lr0 + static_cast<int>(qrow);
            const int col = c0 + static_cast<int>(rcol);
            const double x = Xtab[col], y = Ytab[global_row_of(im, lrow)];
            bool inside = true, near = true;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const double sf = fma(fa[f], x, fma(fb[f], y, fc[f]));
                inside = inside && sf >= 0.0;
                near = near && sf >= -ft[f];
            }
            uint32_t* const at = words + static_cast<size_t>(lrow) * im.res_x + col;
            if (inside) {
                *at = mine;
            } else if (near) {
                const uint32_t old = __hip_atomic_load(at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((old >> 28) != g.split.stamp) atomicCAS(at, old, mine);
            }
        }
    }
}

void launch_plane_raster(hipStream_t s, const GridView& g, const double* Xtab, const double* Ytab, const ImageParams& im) {
    if (g.split.n_slabs <= 1 || g.n_cells <= 0) return;
    // resident wavefronts share the listed cells between them (the list's length is on the device only)
    long long blocks = (g.n_cells + 255) / 256;
    blocks = blocks < 8 ? 8 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(plane_raster, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, g, Xtab, Ytab, im);
}

int64_t walk_tiles(const ImageParams& im) {
    const int64_t tiles_x = (im.res_x + 7) / 8, tiles_y = (im.n_local_rows + 7) / 8;
    return tiles_x * tiles_y;
}

void launch_build_records(hipStream_t s, const GridView& g, double alpha_limit, int order) {
    if (g.n_cells <= 0) return;
    const unsigned blocks = static_cast<unsigned>((g.n_cells + 255) / 256);
    if (g.split.n_slabs > 1 && g.bfrec)
        hipLaunchKernelGGL((build_records<true, true>), dim3(blocks), dim3(256), 0, s, g, alpha_limit, order);
    else if (g.split.n_slabs > 1)
        hipLaunchKernelGGL((build_records<true, false>), dim3(blocks), dim3(256), 0, s, g, alpha_limit, order);
    else if (g.bfrec)
        hipLaunchKernelGGL((build_records<false, true>), dim3(blocks), dim3(256), 0, s, g, alpha_limit, order);
    else
        hipLaunchKernelGGL((build_records<false, false>), dim3(blocks), dim3(256), 0, s, g, alpha_limit, order);
}

void launch_entry_lists(hipStream_t s, const GridView& g, const double* Xtab, const double* Ytab,
                        const ImageParams& im, EntryHead* head, Entry* first, Entry* pool, int64_t capacity,
                        FrameCounters* counters, unsigned* sticky, int want_upper, double key_slack, uint32_t* tile_flag,
                        uint32_t tile_stamp) {
    if (g.n_bfaces <= 0) return;
    const unsigned blocks = static_cast<unsigned>((g.n_bfaces + 3) / 4);
    const RasterArgs A{Xtab, Ytab, im, head, first, pool, capacity, counters, sticky, want_upper, key_slack, tile_flag, tile_stamp,
                       (im.res_x + 7) / 8};
    if (g.bfrec)
        hipLaunchKernelGGL(entry_raster_rec, dim3(blocks), dim3(256), 0, s, g, A);
    else
        hipLaunchKernelGGL(entry_raster, dim3(blocks), dim3(256), 0, s, g, A);
}

void launch_setup_fused(hipStream_t s, const GridView& g, double alpha_limit, int order, const double* Xtab, const double* Ytab,
                        const ImageParams& im, EntryHead* head, Entry* first, Entry* pool, int64_t capacity,
                        FrameCounters* counters, unsigned* sticky, int want_upper, double key_slack) {
    const unsigned n_rec = g.n_cells > 0 ? static_cast<unsigned>((g.n_cells + 255) / 256) : 0u;
    const unsigned n_ras = g.n_bfaces > 0 ? static_cast<unsigned>((g.n_bfaces + 3) / 4) : 0u;
    if (n_rec + n_ras == 0u) return;
    const RasterArgs A{Xtab, Ytab, im, head, first, pool, capacity, counters, sticky, want_upper, key_slack, nullptr, 0u, 0};
    hipLaunchKernelGGL(setup_fused, dim3(n_rec + n_ras), dim3(256), 0, s, g, alpha_limit, order, n_rec, n_ras, A);
}

// ------------------------------------------------------------------------------------------
// walk_composite
// ------------------------------------------------------------------------------------------
// One emission/absorption step in the reference's own arithmetic (line.cpp:220-224):
//   C = Q - alpha * I;   I = (Q - C * exp(-alpha * dz)) / alpha
// with every product and sum rounded separately (no FMA contraction), so that the recurrence —
// including its cancellation noise for tiny alpha — follows the reference's to the last bits of exp.
// The final division is a multiplication by the per-cell reciprocal (<= 1 ulp apart, not amplified).
template <bool kSmallArg = false>
__device__ __forceinline__ double reference_emission_step(double I, double alpha_c, double q, double inv_alpha,
                                                          double dz) {
#pragma clang fp contract(off)
    const double C = q - alpha_c * I;
    const double arg = -alpha_c * dz;
    const double e = kSmallArg ? exp_small_nonpositive(arg) : exp_nonpositive(arg);
    return (q - C * e) * inv_alpha;
}

// Geometry of one step: how far along the walk the ray (x, y) leaves the current cell, and through which face.
// The record holds the (up to three) faces a ray can leave through as planes of the walk coordinate w about the
// absolute pixel coordinates (device_types.hpp: ExitRecord); the chord is w_exit minus the depth at which the ray
// entered — the exit depth of the cell before, or the boundary entry's own depth (the caller's `carry`).
struct StepGeometry {
    double w_exit;     // min over the candidates; +inf: the ray cannot leave (flat cell, edge-on faces): it ends here
    uint32_t w_out;    // neighbour word of the exit face
};

// ORDER 0: walk from -z to +z and integrate back to front exactly like
//          line::integrate_ray_value_by_i (the reference sorts by z_hi descending and runs the
//          recurrence from the last element, line.cpp:138,206).  Default: parity first.
// ORDER 1: walk from +z (the viewer) to -z, I = sum_k T_k S_k with the transmittance early-out
//          (wavefront-uniform skip of the exp work once every lane's T fell below the cut-off).
//          Algebraically identical; differs from ORDER 0 by rounding only where the reference's
//          recurrence is itself well conditioned.
//
// Structure of a step (one cell of one ray): the record of the current cell is already in
// registers; its three candidate planes give the exit face and therefore the next cell; the loads of
// the NEXT cell's record are issued right there, and the exp/divide work of the CURRENT cell runs
// while they are in flight.  Face selection is branch-free (selects), so a step has two
// data-dependent branches only: "this cell contributes" and "the ray left the grid".
struct CellRegs {
    D2 r0, r1, r2, r3, r4, r5, r6, r7;  // ExitRecord: r0-r4 planes (+ nbr[0..1] in r4.b), r5.a = nbr[2] | flags, r6 / r7 optics
};

__device__ __forceinline__ void load_cell(CellRegs& c, const ExitRecord* rec, int cell) {
    const D2* rp = reinterpret_cast<const D2*>(rec + cell);
    c.r0 = rp[0];
    c.r1 = rp[1];
    c.r2 = rp[2];
    c.r3 = rp[3];
    c.r4 = rp[4];
    c.r5 = rp[5];
    c.r6 = rp[6];
    c.r7 = rp[7];
}

// min as ONE instruction (fmin() on a value the compiler cannot prove canonical is preceded by a canonicalising
// v_max_f64 x, x; the operands here are never NaN: a candidate is a finite depth or +inf)
__device__ __forceinline__ double min_f64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ StepGeometry step_geometry(const CellRegs& cur, double x, double y) {
    // plane k = (c, gx, gy): r0.a r0.b r1.a | r1.b r2.a r2.b | r3.a r3.b r4.a
    const double w0 = fma(cur.r0.b, x, fma(cur.r1.a, y, cur.r0.a));
    const double w1 = fma(cur.r2.a, x, fma(cur.r2.b, y, cur.r1.b));
    const double w2 = fma(cur.r3.b, x, fma(cur.r4.a, y, cur.r3.a));
    const unsigned long long n01 = __double_as_longlong(cur.r4.b);
    const uint32_t n0 = static_cast<uint32_t>(n01), n1 = static_cast<uint32_t>(n01 >> 32);
    const uint32_t n2 = static_cast<uint32_t>(__double_as_longlong(cur.r5.a));
    StepGeometry g;
    g.w_exit = min_f64(w0, min_f64(w1, w2));
    g.w_out = (w0 == g.w_exit) ? n0 : (w1 == g.w_exit) ? n1 : n2;
    return g;
}

template <int TILE, int ORDER>
__global__ __launch_bounds__(256) void walk_composite(WalkParams P) {
    using TS = TileShape<TILE>;
    constexpr int TW = TS::WW * TS::GX, TH = TS::WH * TS::GY;
    constexpr bool kUp = (ORDER == 0);  // walking towards +z
    const ImageParams& im = P.im;
    const int tiles_x = (im.res_x + TW - 1) / TW;
    const int tiles_y = (im.n_local_rows + TH - 1) / TH;

    int tx, ty;
    if (P.xcd_mode == 0) {
        ty = blockIdx.x / tiles_x;
        tx = blockIdx.x - ty * tiles_x;
    } else {
        // Bands of up to 32 image rows are dealt round-robin to the 8 XCDs (blocks b and b + 8 share an
        // XCD's L2): every XCD sweeps the image top to bottom, so load stays balanced, while the
        // workgroups resident on one XCD at a time cover a compact region of the grid.
        const int BAND = P.band_tiles;  // tile rows per band (host: ~32 image rows, fewer for short strips)
        const int n_bands = (tiles_y + BAND - 1) / BAND;
        const int per_band = BAND * tiles_x;
        const int xcd = blockIdx.x & 7;
        const int seq = blockIdx.x >> 3;
        const int band = (seq / per_band) * 8 + xcd;
        const int within = seq - (seq / per_band) * per_band;
        if (band >= n_bands) return;
        ty = band * BAND + within % BAND;
        tx = within / BAND;
        if (ty >= tiles_y) return;
    }

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = tx * TW + (wave % TS::GX) * TS::WW + (lane % TS::WW);
    const int lrow = ty * TH + (wave / TS::GX) * TS::WH + (lane / TS::WW);
    const bool in_image = (col < im.res_x) && (lrow < im.n_local_rows);

    unsigned n_seg = 0, n_step = 0, is_solid = 0, overflow = 0;
    bool skipped = false;        // the ray met an entry inside a stretch it had walked (next_entry): interpenetrating components
    double key_taken = -DBL_MAX; // key of the entry the ray took last
    double tau = 0.0, I = 0.0, T = 1.0;
    double x = 0.0, y = 0.0, w_cur = -DBL_MAX, carry = 0.0;  // carry: the depth (walk coordinate) at which the ray entered the current cell
    EntryHead ent{0, 0};
    int cell = -1;
    size_t lp = 0;
    float2 result = make_float2(0.f, 0.f);

    if (in_image) {
        lp = static_cast<size_t>(lrow) * im.res_x + col;
        const uint32_t mv = P.mask ? P.mask[lp] : 0u;
        if (mv) {
            // line.cpp:177-179,197-199: a solid-marked pixel returns the mark on both channels
            double colour = 0.0;
            for (int s = 0; s < P.solids.n_slots; ++s)
                if (mv == static_cast<uint32_t>(s) + 1u) colour = P.solids.colour[s];
            result.x = static_cast<float>(colour);
            result.y = result.x;
            is_solid = 1;
        } else {
            x = P.Xtab[col];
            y = P.Ytab[global_row_of(im, lrow)];
            // touched once per frame: keep them from displacing the cell records in L2 / Infinity Cache
            ent = load_entry_head(P.entry_head + lp);
            if (ent.count > 0) cell = next_entry<kUp>(P, lp, ent, w_cur, carry, -DBL_MAX, -DBL_MAX, skipped);
            key_taken = w_cur;
        }
    }

    CellRegs cur;
    if (cell >= 0) load_cell(cur, P.xrec, cell);

    while (cell >= 0) {
        const StepGeometry sg = step_geometry(cur, x, y);
        ++n_step;
        const bool has_exit = sg.w_exit < INFINITY;
        const double dz = sg.w_exit - carry;  // line.cpp:124-131: the chord through the cell
        const bool contributes = dz > 0.0 && dz < INFINITY;

        // where next?
        int nb = -1;
        double carry_next = carry;
        if (has_exit) {
            carry_next = sg.w_exit;
            w_cur = fmax(w_cur, sg.w_exit);
            const uint32_t id = sg.w_out & kIdMask;
            if (id != kNoCell) nb = static_cast<int>(id);
        }
        if (nb >= 0 && n_step >= P.max_steps) {  // malformed grid: never spin
            overflow = 1;
            nb = -1;
        } else if (nb < 0 && !overflow) {
            // left the grid: re-entry of a non-convex grid?
            nb = next_entry<kUp>(P, lp, ent, w_cur, carry_next, key_taken, has_exit ? sg.w_exit : -DBL_MAX, skipped);
            key_taken = w_cur;
        }

        // issue the next cell's loads now; the arithmetic below does not depend on them
        CellRegs nxt;
        if (nb >= 0) load_cell(nxt, P.xrec, nb);

        if (contributes) {
            ++n_seg;
            tau = fma(dz, cur.r6.a, tau);  // line.cpp:189 (unclamped alpha)
            if (ORDER == 0) {
                // line.cpp:220-224 (NaN alpha propagates like there)
                if (cur.r6.b != 0.0) I = reference_emission_step(I, cur.r6.b, cur.r7.b, cur.r7.a, dz);
            } else if (T >= P.t_cutoff) {
                // I = sum_k T_k (Q/alpha)(1 - e^{-alpha dz}); T_{k+1} = T_k e^{-alpha dz}
                const double ex = exp_nonpositive(-cur.r6.b * dz);
                I = fma(T * cur.r7.a, 1.0 - ex, I);
                T *= ex;
            }
        }
        cell = nb;
        carry = carry_next;
        cur = nxt;
    }

    if (in_image) {
        if (!is_solid) {
            result.x = static_cast<float>(tau);  // plane.cpp:165
            result.y = static_cast<float>(I);    // plane.cpp:166
        }
        if (!P.keep_entries) __builtin_nontemporal_store(0ll, reinterpret_cast<long long*>(P.entry_head + lp));  // cleared for the next frame
        __builtin_nontemporal_store(result.x, &P.out[lp].x);
        __builtin_nontemporal_store(result.y, &P.out[lp].y);
    }

    // segments per image row (load balancing of row blocks across GPUs): reduce over the lanes of a
    // wavefront that share a row (WW consecutive lanes), one atomic per row and wavefront
    if (P.row_cost) {
        unsigned rs = n_seg;
#pragma unroll
        for (int d = TS::WW / 2; d >= 1; d >>= 1) rs += __shfl_xor(rs, d);
        if ((lane % TS::WW) == 0 && rs && lrow < im.n_local_rows) atomicAdd(P.row_cost + lrow, rs);
    }

    // per-wavefront statistics -> one atomic each
    const unsigned s_seg = wave_sum_u32(n_seg);
    const unsigned s_step = wave_sum_u32(n_step);
    const unsigned s_cov = wave_sum_u32(n_seg > 0 ? 1u : 0u);
    const unsigned s_sol = wave_sum_u32(is_solid);
    const unsigned s_ovf = wave_sum_u32(overflow);
    const unsigned s_ent = wave_sum_u32(static_cast<unsigned>(ent.count));
    const unsigned s_skip = static_cast<unsigned>(__popcll(__builtin_amdgcn_ballot_w64(skipped)));
    if (lane == 0) {
        if (s_skip) {
            atomicAdd(&P.counters->overlap_rays, s_skip);
            atomicAdd(P.sticky + 2, s_skip);
        }
        FrameCounters* const fc = P.counters + ((blockIdx.x * 4u + static_cast<unsigned>(threadIdx.x >> 6)) % kCounterShards);
        if (s_seg) atomicAdd(&fc->seg_tiles, static_cast<unsigned long long>(s_seg) | (1ull << kCounterHighShift));
        if (s_step | s_cov) atomicAdd(&fc->steps_cov, static_cast<unsigned long long>(s_step) | (static_cast<unsigned long long>(s_cov) << kCounterHighShift));
        if (s_ent | s_sol) atomicAdd(&fc->ent_solid, static_cast<unsigned long long>(s_ent) | (static_cast<unsigned long long>(s_sol) << kCounterHighShift));
        if (s_ovf) {  // (shard 0, beside entry_overflow: the two words a frame's status is read from; rare, never contended)
            atomicAdd(&P.counters->walk_overflow, s_ovf);
            atomicAdd(P.sticky + 1, s_ovf);
        }
    }
}

// ------------------------------------------------------------------------------------------
// walk_composite_lds: the same walk with the current cells staged through LDS.
//
// Neighbouring pixels of a row tile are mostly inside the same cell, so per step a wavefront
// needs only a handful of distinct 128-byte records, not 64.  The direct kernel still issues
// eight 16-byte loads per lane and step and is bound by the CU's vector-memory address path.
// Here the wavefront
//   1. gives every DISTINCT next-cell id among its lanes a slot (leader election, below),
//   2. loads each slot's record ONCE, cooperatively: 8 lanes x 16 B per ExitRecord (one 128-byte line),
//   3. parks the pieces in a wave-private LDS area (144-byte stride: conflict-free b128 reads),
//   4. and every lane reads its own cell's record from LDS (lanes of a run broadcast).
// The global loads of step k+1 are issued as soon as the exit face of step k is known and are in
// flight during step k's exp/divide work, as in the direct kernel.
// ------------------------------------------------------------------------------------------
#ifndef C5_ELECT_LEADERS
#define C5_ELECT_LEADERS 1
#endif
constexpr bool kElectLeaders = C5_ELECT_LEADERS != 0;  // 0: slots per run of equal ids along the lanes (the older scheme)
// Measured on the C3 frame at 8x8 tiles (7 distinct cells per step on average): 16 slots / 6 wavefronts
// per SIMD 0.663 ms; 32 slots 0.753 (LDS then caps the CU at 5 workgroups); 8 slots 0.735 (direct-load
// fallback too often); 7 wavefronts per SIMD (72 VGPRs, 14 spilled) 0.662.
#ifndef C5_STAGE_SLOTS
#define C5_STAGE_SLOTS 14
#endif
#ifndef C5_WALK_WAVES
#define C5_WALK_WAVES 6
#endif
constexpr int kStageSlots = C5_STAGE_SLOTS;
#ifndef C5_ELECT_BUCKETS
#define C5_ELECT_BUCKETS 256
#endif
constexpr unsigned kBuckets1 = C5_ELECT_BUCKETS;  // first leader table (power of two)   // runs of equal cell ids staged per wavefront and step (more: direct loads)
// One staged cell in LDS, 16-byte units: the 8 of its ExitRecord + 1 pad.  144 bytes = 36 banks: sixteen consecutive
// slots start on sixteen different 4-bank columns (slots s and s + 16 share theirs), so the lanes of a 16-lane group
// read their cells without bank conflicts; a DMA pass (64 lanes, destinations lane-linear) stages seven whole slots,
// its pad lanes idle.  (A stride of 8 units would let a pass stage eight, at the price of two-way conflicts between
// slots of equal parity.)
#ifndef C5_SLOT_STRIDE
#define C5_SLOT_STRIDE 9
#endif
constexpr int kSlotStride = C5_SLOT_STRIDE;
using V2 = double __attribute__((ext_vector_type(2)));  // 16 bytes as one SSA value (never an alloca)
__device__ __forceinline__ D2 as_d2(V2 v) { return D2{v.x, v.y}; }

// Optional in-kernel phase clock (build with -DC5_WALK_STAMPS=1; scripts/stamp_walk.py): ONE WAVEFRONT IN 67 (every XCD in turn) sums, per
// phase of a step, the shader cycles between stamps (s_memtime; tick = shader cycle) and lane 0 adds them to
// g_walk_stamps; the other 63 run the product's instruction stream, so that the sampled wavefronts see the memory
// system, the LDS and the issue ports as loaded as the product's do.  (Stamping every wavefront — the round-2 build —
// made a step five times slower: 28 wavefronts per CU queueing for s_memtime five times per step, and 14 same-address
// atomics per wavefront at the end of the launch.)
#ifndef C5_WALK_STAMPS
#define C5_WALK_STAMPS 0
#endif
#if C5_WALK_STAMPS
__device__ unsigned long long g_walk_stamps[16];
// launch timeline: per workgroup {start, end} in s_memrealtime ticks (100 MHz, one clock for the whole chip) and
// {XCC_ID | HW_ID << 8, wavefront-steps}; zero = the workgroup had no ray (scripts/walk_timeline.py)
__device__ unsigned long long g_walk_trace[4 * 131072];
#define C5_STAMP(k)                                                      \
    do {                                                                 \
        if (stamping_) {                                                 \
            const unsigned long long t_now_ = __builtin_amdgcn_s_memtime(); \
            stamp_acc[k] += t_now_ - t_prev_;                            \
            t_prev_ = t_now_;                                            \
        }                                                                \
    } while (0)
#elif defined(C5_ISA_MARKERS)
// scripts/walk_isa.py: the phase boundaries of the phase clock as comments in the assembly of the PRODUCT's kernel
// (an empty asm statement: no instruction; the script checks that the loop's instruction count is the product's)
#define C5_STAMP(k) asm volatile("; C5_PHASE_END " #k)
#else
#define C5_STAMP(k) \
    do {            \
    } while (0)
#endif

// DMA = true (option "lds_stage" 2): the staging loads write LDS themselves (global_load_lds_dwordx4: destination =
// wave-uniform base + 16 * lane, no vector register in between, no ds_write_b128 — a 16-byte LDS store costs 13
// LDS cycles per wave-instruction, a step's three as much as its ten reads).  A pass of 64 lanes fills 64
// consecutive 16-byte units of the wavefront's staging area: seven whole slots of nine units (eight pieces of the
// cell's one 128-byte line + a pad unit whose lane idles), the same (slot within the pass, piece) for a lane in
// every pass, so three registers describe its part: where that slot's cell id is posted, and the piece's offset.
// Wavefronts per SIMD of the LDS-DMA kernel: 8 (62 VGPRs with the emission integrated at once, below).  Measured on the
// exit-record kernel (C3 frame / C3 at 4800x3600 / C2 ball, walk ms): 7 wavefronts with the emission deferred into
// the next step's load shadow 0.4942 / 1.671 / 0.0965; emission at once, 7 wavefronts 0.4954 / 1.678 / 0.0956;
// emission at once, 8 wavefronts 0.4786 / 1.618 / 0.0944; deferred, 8 wavefronts (64 VGPRs, 12 B of scratch)
// 0.4832 / 1.622 / 0.1043.  (On the four-plane records of rounds 1-2 an eighth wavefront bought nothing: the vector
// issue was full at seven; the exit records issue a fifth fewer vector instructions per step.)
#ifndef C5_DMA_WAVES
#define C5_DMA_WAVES 8
#endif
// 1 (DMA kernel only): a step's emission/absorption is integrated at once, behind its geometry — nothing carried to
// the next iteration, ten registers fewer — instead of in the shadow of the next step's loads
#ifndef C5_EMIT_NOW
#define C5_EMIT_NOW 1
#endif
// 1: the scalar trims of round 4 in the loop of walk_composite_lds (profiles/r04_walk_isa.md); 0: the loop as it was (A/B builds)
#ifndef C5_LOOP_TRIM
#define C5_LOOP_TRIM 1
#endif
using LdsInts = const __attribute__((address_space(3))) int*;
// SLOTS: distinct cells staged per wavefront and step: 14 (two DMA passes of seven), or 21 (three) for frames whose
// pixels are coarse against the cells: more distinct cells per 8x8 tile — the host picks by the rays per cell of the
// frame before (c_api.hip).
// SMALLEXP: every exp argument of the frame lies in (-1/8, 0] (WalkParams::small_exp_only): only the short series.
// SPLIT ("depth_split", device_types.hpp: SplitParams): every tile is walked by P.split.n_slabs jobs, one per slab of
// depth; a job leaves partial results and the tile's last job to arrive composes them.  ORDER 0, one wavefront per
// workgroup.  Everything it adds is compiled out of the whole-ray instantiations.
template <int TILE, int ORDER, bool DMA = false, int SLOTS = kStageSlots, bool SMALLEXP = false, bool SPLIT = false>
__global__ __launch_bounds__(256, DMA ? ((SLOTS > 16 || SPLIT) ? 7 : C5_DMA_WAVES) : C5_WALK_WAVES) void walk_composite_lds(WalkParams P) {
    static_assert(!SPLIT || (ORDER == 0 && DMA && TileShape<TILE>::GX * TileShape<TILE>::GY == 1), "depth_split: reference order, LDS-DMA, one wavefront per workgroup");
    constexpr int kStageSlots = SLOTS;  // (shadows the namespace constant: everything below is per instantiation)
    using TS = TileShape<TILE>;
    constexpr int TW = TS::WW * TS::GX, TH = TS::WH * TS::GY;
    constexpr bool kUp = (ORDER == 0);
    constexpr bool kEmitNow = DMA && C5_EMIT_NOW != 0;
    constexpr int kWaves = TileShape<TILE>::GX * TileShape<TILE>::GY;  // wavefronts per workgroup
    __shared__ V2 s_stage[kWaves][kStageSlots * kSlotStride];
    // per wavefront: leader tables of kBuckets1 and 64 buckets + the cell id of every slot
    __shared__ int s_elect[kWaves][kBuckets1 + 128];
    __shared__ double s_scur[kWaves][64];

    const ImageParams& im = P.im;
    const int tiles_x = (im.res_x + TW - 1) / TW;
    const int tiles_y = (im.n_local_rows + TH - 1) / TH;
    // SPLIT: the K jobs of a tile follow one another on the same XCD (blocks b and b + 8 share one): block -> (tile block, slab)
    int slab = 0;
    unsigned bid = blockIdx.x;
    if (SPLIT) {
        const unsigned K = static_cast<unsigned>(P.split.n_slabs);
        if (P.xcd_mode == 0) {
            slab = static_cast<int>(bid % K);
            bid = bid / K;
        } else {
            const unsigned seq_k = bid >> 3;
            slab = static_cast<int>(seq_k % K);
            bid = ((seq_k / K) << 3) | (bid & 7u);
        }
    }
    int tx, ty;
    if (P.xcd_mode == 0) {
        ty = bid / tiles_x;
        tx = bid - ty * tiles_x;
    } else if (P.xcd_mode == 2) {
        // square super-blocks of S x S workgroups dealt round-robin to the 8 XCDs: the workgroups an XCD
        // runs at a time are neighbours in x AND y, so a cell's record is fetched into few L2s
        const int S = P.band_tiles;
        const int sbx_n = (tiles_x + S - 1) / S, sby_n = (tiles_y + S - 1) / S;
        const int xcd = bid & 7;
        const int seq = bid >> 3;
        const int sb = (seq / (S * S)) * 8 + xcd;
        const int within = seq - (seq / (S * S)) * (S * S);
        if (sb >= sbx_n * sby_n) return;
        int sby = sb / sbx_n;
        const int sbx = sb - sby * sbx_n;
        if (P.n_sb_rows == sby_n) sby = static_cast<int>(P.sb_order[sby]);  // rows with the longest rays first (kernel argument)
        ty = sby * S + within / S;
        tx = sbx * S + (within - (within / S) * S);
        if (tx >= tiles_x || ty >= tiles_y) return;
    } else {
        const int BAND = P.band_tiles;
        const int n_bands = (tiles_y + BAND - 1) / BAND;
        const int per_band = BAND * tiles_x;
        const int xcd = bid & 7;
        const int seq = bid >> 3;
        const int band = (seq / per_band) * 8 + xcd;
        const int within = seq - (seq / per_band) * per_band;
        if (band >= n_bands) return;
        ty = band * BAND + within % BAND;
        tx = within / BAND;
        if (ty >= tiles_y) return;
    }
    // SPLIT: this job's slab of depth [w_lo, w_hi) at the lane's pixel (the first from -DBL_MAX, the last to +DBL_MAX; the
    // planes share a tilt: device_types.hpp); w_lo is wanted at the start and at re-entries only and is worked out there
    double w_hi = DBL_MAX;
    auto slab_lo = [&](double px, double py) { return SPLIT ? P.split.w[slab] + fma(P.split.gx, px, P.split.gy * py) : -DBL_MAX; };

    // Registers are what caps the resident wavefronts here, so per-lane state the steps do not need
    // (pixel index, entry head, solid colour, s_cur) is NOT carried through the loop: it is recomputed
    // or re-read on the rare paths that want it (start, re-entry, end).
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform: LDS bases stay scalar
    auto pixel_col = [&]() { return tx * TW + (wave % TS::GX) * TS::WW + (lane % TS::WW); };
    auto pixel_lrow = [&]() { return ty * TH + (wave / TS::GX) * TS::WH + (lane / TS::WW); };
    auto pixel_index = [&]() { return static_cast<size_t>(pixel_lrow()) * im.res_x + pixel_col(); };
    // the same from a lane index the compiler cannot trace back to `lane`: on the rare paths inside and behind the loop the
    // pixel's index is then COMPUTED there (three instructions) instead of being hoisted in front of the loop and spilled
    auto pixel_index_rare = [&]() {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int c = tx * TW + (wave % TS::GX) * TS::WW + (l % TS::WW), r = ty * TH + (wave / TS::GX) * TS::WH + (l / TS::WW);
        return static_cast<size_t>(r) * im.res_x + c;
    };
    const bool in_image = (pixel_col() < im.res_x) && (pixel_lrow() < im.n_local_rows);
    V2* const my_stage = s_stage[wave];
    int* const my_elect = s_elect[wave];
    double* const my_scur = s_scur[wave];  // s_cur of every lane: only read and written around (re-)entries
    // records and optics are addressed as a uniform base + a 32-bit byte offset per lane (the host
    // only picks this kernel while n_cells * 128 fits 32 bits): one shift-or per load instead of a
    // 64-bit shift and a 64-bit add
    const char* const rec_bytes = reinterpret_cast<const char*>(P.xrec);

    constexpr unsigned kOverflowBit = 0x80000000u;  // of n_seg: the ray hit the step bound
    constexpr unsigned kSkippedBit = 0x40000000u;   // of n_seg: the ray met an entry inside a stretch it had walked (next_entry)
    constexpr unsigned kClippedBit = 0x20000000u;   // of n_seg (SPLIT): the ray goes on beyond the job's slab
    unsigned n_seg = 0;
    unsigned n_step_wave = 0;  // wave-uniform: lane-steps taken by the whole wavefront
    double tau = 0.0, I = 0.0, T = 1.0;
    double tauc = 0.0;  // SPLIT: sum of dz * clamped alpha over the job's cells: exp(-tauc) is what the job does to the I below it
    double x = 0.0, y = 0.0;
    double carry = 0.0;  // the depth (walk coordinate) at which the ray entered the cell it is about to cross
    int nb = -1;

    {
        // Most wavefronts of a frame see neither the grid nor a solid (70 % on the C3 frame): they store
        // their zeros and leave at once (their entry heads are zero already).  An empty pixel used to
        // cost as much as six ray-cell segments.
        size_t lp = 0;
        uint32_t mv = 0;
        EntryHead ent{0, 0};
        // (round 4) The raster marks the tiles that hold an entry (RasterArgs::tile_flag).  One scalar load says whether this
        // one does: if not, the zeros are stored without reading 64 entry heads; if so, heads, first entries and pixel
        // coordinates are asked for in ONE round of loads instead of two dependent ones.  (Default tile shape; no solids.)
        const bool flagged = (TILE == 3) && P.tile_flag != nullptr;
        bool tile_has_entries = true;
        if (flagged) tile_has_entries = P.tile_flag[static_cast<size_t>(ty) * tiles_x + tx] == P.tile_stamp;  // (uniform address)
        double first_z = 0.0;
        uint32_t first_cell = 0u;
        if (in_image) {
            lp = pixel_index();
            mv = P.mask ? P.mask[lp] : 0u;
            if (tile_has_entries) {
                // touched once per frame: keep them from displacing the cell records in L2 / Infinity Cache
                ent = load_entry_head(P.entry_head + lp);
                if (flagged) {
                    const Entry first_ahead = P.entry_first[lp];
                    first_z = first_ahead.z;
                    first_cell = first_ahead.cell;
                    x = P.Xtab[pixel_col()];
                    y = P.Ytab[global_row_of(im, pixel_lrow())];
                }
            }
        }
        if (__builtin_amdgcn_ballot_w64(mv != 0u || ent.count != 0) == 0ull) {
            // (SPLIT: every job of the tile comes to the same verdict; the first one writes the zeros)
            if (in_image && slab == 0) {
                __builtin_nontemporal_store(0.f, &P.out[lp].x);
                __builtin_nontemporal_store(0.f, &P.out[lp].y);
            }
            return;
        }
        if (in_image && !mv) {  // (a solid-marked pixel is written at the end, without a walk)
            if (!flagged) {
                x = P.Xtab[pixel_col()];
                y = P.Ytab[global_row_of(im, pixel_lrow())];
            }
            double w_cur = -DBL_MAX;
            bool skipped = false;
            const double w_lo = slab_lo(x, y);
            if (SPLIT) w_hi = P.split.w[slab + 1] + fma(P.split.gx, x, P.split.gy * y);
            if (SPLIT && slab > 0) {
                // where is the ray at the cutting plane?  (plane_raster; a word of another frame: nowhere inside the grid)
                const uint32_t pc = P.split.plane_cell[static_cast<size_t>(slab - 1) * P.split.plane_stride + lp];
                if ((pc >> 28) == P.split.stamp) {
                    nb = static_cast<int>(pc & kIdMask);
                    carry = w_lo;
                }
            }
            if (nb < 0 && ent.count > 0)
                nb = next_entry<kUp>(P, lp, ent, w_cur, carry, -DBL_MAX, -DBL_MAX, skipped, w_lo, w_hi, flagged, first_z, first_cell);
            my_scur[lane] = w_cur;  // the key of the entry taken (-DBL_MAX: started from a plane)
        }
    }
    my_elect[kBuckets1 + 64 + lane] = 0;  // slot ids: always a valid cell id, whatever the slot's state
    // (recomputed where it is wanted: nothing to keep in a register across the loop)
    auto fc_job = [&]() { return P.counters + ((blockIdx.x * 4u + static_cast<unsigned>(wave)) % kCounterShards); };
    // one tile in sixteen (the first of its super-block; 1 x 1 where the launch has no super-blocks) speaks for the estimates
    const bool sampler = P.xcd_mode != 2 || ((tx % P.band_tiles) == 0 && (ty % P.band_tiles) == 0);
    {   // the shallowest depth at which a ray of this job starts (c_api.hip places the next frame's cutting planes by it)
        double lo = nb >= 0 ? carry : DBL_MAX;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) lo = fmin(lo, __shfl_xor(lo, d));
        if (lane == 0 && lo < DBL_MAX && sampler) atomicMax(&fc_job()->entry_min_key, depth_key(-lo));
    }

    // contribution of the step whose record is being replaced (integrated while the next loads fly)
    // (optics kept as the two 16-byte halves they are read in: {alpha_raw, alpha_c}, {aux, q})
    bool pend = false;
    double pend_dz = 0.0;
    V2 pend_o0 = {0.0, 0.0}, pend_o1 = {0.0, 0.0};

    // lane-constant pieces of the staging addresses
    static_assert(kStageSlots <= 32, "slot ids live in 64 ints; the register path stages passes of 8 slots");
    constexpr int kPasses = (kStageSlots + 7) / 8;  // register path: 8 slots (8 lanes x 16 B each) per pass
    const int piece = lane & 7, sub = lane >> 3;
    const uint32_t rec_piece_off = static_cast<uint32_t>(piece) * 16u;
    const int sub4 = sub << 2;  // ds_bpermute byte address of slot `sub`
    V2* const put_rec = my_stage + sub * kSlotStride + piece;         // + 8 * pass * kSlotStride
    // DMA: a pass stages kDmaSlots = 7 whole slots (63 lanes, of which the seven pad lanes and lane 63 idle), so that a
    // lane fetches the same piece of the same slot-within-the-pass in every pass: two registers hold where that
    // slot's cell id is posted and the piece's offset inside the cell's 128-byte line.
    // Pass j: slots 7 j ... 7 j + 6, LDS units from 63 j.
    constexpr int kDmaSlots = 64 / kSlotStride;
    constexpr int kDmaPasses = (kStageSlots + kDmaSlots - 1) / kDmaSlots;
    uint32_t dma_id_at = 0, dma_off = 0;
    if (DMA) {
        const uint32_t ids_at = (uint32_t)(uintptr_t)(LdsInts)(my_elect + kBuckets1 + 64);  // LDS byte address
        const int s_ = lane / kSlotStride, pc = lane - s_ * kSlotStride;
        const bool has = s_ < kDmaSlots && pc < 8;
        // (idle lanes: a slot no pass ever stages — still an address inside the workgroup's LDS, so that the id reads
        // below need no predicate and are all in flight before the first load is issued)
        dma_id_at = ids_at + 4u * static_cast<uint32_t>(has ? s_ : 60);
        dma_off = 16u * static_cast<uint32_t>(pc & 7);
    }

    // A lane steps in every iteration from its start to its end (re-entry takes no extra iteration), so
    // the wave-uniform iteration count IS the step count of every lane still walking: the guard
    // against malformed grids (never spin) is one scalar compare per iteration.
#if C5_WALK_STAMPS
    const bool stamping_ = (blockIdx.x % 67u) == 5u;  // wave-uniform; a stride that visits every XCD (blockIdx & 7) in turn
    const unsigned long long trace_begin_ = __builtin_amdgcn_s_memrealtime();
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned stat_runs = 0, stat_distinct = 0, stat_iters = 0, stat_lanes = 0;
#if C5_WALK_STAMPS > 1
    unsigned stat_next_staged = 0, stat_next_total = 0;
    int nb_before_ = -1;
#endif
    unsigned long long t_prev_ = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin_ = t_prev_;
#endif
    // (the bound against malformed grids, made opaque: read straight from the kernel arguments the compiler re-loaded it -
    // a scalar memory load and its wait - at the head of EVERY step rather than keep it in a register)
    uint32_t max_steps = P.max_steps;
    asm volatile("" : "+s"(max_steps));
    for (unsigned iter = 0;; ++iter) {
        const bool need = nb >= 0;
        const unsigned long long needs = __builtin_amdgcn_ballot_w64(need);
        if (needs == 0ull) break;
        if (iter >= (C5_LOOP_TRIM ? max_steps : P.max_steps)) {
            if (need) n_seg |= kOverflowBit;
            break;
        }
        n_step_wave += static_cast<unsigned>(__popcll(needs));

        // 1. lanes in the same cell -> one slot.
        int n_runs, slot;
        if (kElectLeaders) {
            // Neighbouring rays drift out of phase, so equal ids are rarely ADJACENT along the lanes: the C3
            // frame has 22-27 runs of equal ids per step but only 7-17 distinct cells (8 in a 16x4 tile).
            // Elect one leader per distinct cell through a table of kBuckets1 buckets in LDS (7 cells share
            // a bucket in one step of twelve at 256 buckets, in one of three at 64): every walking lane
            // writes (id << 6 | lane) to bucket hash(id) — ids of this kernel have 25 bits — and reads the
            // bucket's winner back: one LDS round trip tells it the winner's lane AND whether the winner
            // is in the same cell.  All lanes of a cell hash alike, so they follow the winner together or
            // stay unresolved together (bucket shared with another cell); the unresolved go through a
            // second table with another hash, and whoever is left after that leads itself.
            const unsigned unb = static_cast<unsigned>(nb);
            const unsigned h1 = (unb ^ (unb >> 8)) & (kBuckets1 - 1u);
            const int ticket = static_cast<int>((unb << 6) | static_cast<unsigned>(lane));
            if (need) my_elect[h1] = ticket;
            __builtin_amdgcn_wave_barrier();
            const int won = my_elect[h1];  // lanes without a ray read some old ticket: harmless, they match nothing
            __builtin_amdgcn_wave_barrier();
            int w = won & 63;
            const bool other = (won >> 6) != nb;
            const bool open = need && other;
            if ((__builtin_amdgcn_ballot_w64(other) & needs) != 0ull) {  // wave-uniform; about one step in twelve
                const unsigned t = unb >> 6;
                const unsigned h2 = (unb + t + (t << 2) + (unb >> 12)) & 63u;
                if (open) my_elect[kBuckets1 + h2] = ticket;
                __builtin_amdgcn_wave_barrier();
                const int won2 = my_elect[kBuckets1 + h2];
                __builtin_amdgcn_wave_barrier();
                if (open) w = ((won2 >> 6) == nb) ? (won2 & 63) : lane;
            }
            // (uicmp: the compare straight into a lane mask; ballot() of this bool went through a 0/1 register)
            const unsigned long long heads =
                __builtin_amdgcn_uicmp(static_cast<unsigned>(w), static_cast<unsigned>(lane), 32 /* eq */) & needs;
            n_runs = __builtin_popcountll(heads);
            // a leader's slot = leaders below it.  The leaders post their cell id for the loader lanes
            // (my_elect[kBuckets1 + 64 + slot]; slots not in use keep an older, still valid id) while everybody
            // fetches its leader's slot.
            const int rank = static_cast<int>(__builtin_amdgcn_mbcnt_hi(
                static_cast<uint32_t>(heads >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(heads), 0u)));
            if (need && w == lane) my_elect[kBuckets1 + 64 + rank] = nb;
            slot = __builtin_amdgcn_ds_bpermute(w << 2, rank);
            __builtin_amdgcn_wave_barrier();
        } else {
            // runs of equal ids along the lanes.  `left` = nb of the lane to the left (DPP wave shift, one
            // VALU instruction, no LDS round trip); lane 0 reads 0 and is a head by decree.
            const int left = __builtin_amdgcn_mov_dpp(nb, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
            const unsigned long long heads = (__builtin_amdgcn_ballot_w64(left != nb) | 1ull) & needs;
            n_runs = __popcll(heads);
            // slot = (heads at or below this lane) - 1, as mbcnt over heads >> 1 seeded with (bit 0) - 1
            const unsigned long long hs = heads >> 1;
            slot = static_cast<int>(__builtin_amdgcn_mbcnt_hi(
                static_cast<uint32_t>(hs >> 32),
                __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(hs), static_cast<uint32_t>(static_cast<int>(heads & 1ull) - 1))));
        }
        // (kept in a scalar register by hand: as a plain min() the compiler compared it on the vector unit)
        const int n_runs_s = __builtin_amdgcn_readfirstlane(n_runs);  // (the count of a ballot: uniform; said so, it is compared on the scalar unit)
        const int n_staged = n_runs_s < kStageSlots ? n_runs_s : kStageSlots;
#if C5_WALK_STAMPS
        stat_iters += 1u;
        stat_lanes += static_cast<unsigned>(__popcll(needs));
#endif
#if C5_WALK_STAMPS > 1
        {   // coherence statistics (expensive: -DC5_WALK_STAMPS=2): runs of equal ids against distinct ids among the walking lanes
            unsigned long long seen = 0ull;  // lanes whose id already occurred in a lower lane
            for (int l = 0; l < 64; ++l) {
                const int v = __builtin_amdgcn_readlane(nb, l);
                if (v < 0) continue;
                seen |= __builtin_amdgcn_ballot_w64(nb == v && lane > l);
            }
            stat_runs += static_cast<unsigned>(n_runs);
            stat_distinct += static_cast<unsigned>(__popcll(needs & ~seen));
        }
#endif
        // lane s learns the cell id of slot s: every lane of a group pushes the same id to lane `slot`
        // (finished rays push to lane 63, which is a slot only when all 64 lanes are leaders).  Lanes
        // nobody pushes to read 0 (ds_permute_b32 clears its buffer first; probed on gfx950,
        // scripts/probes/lane_ops_probe.hip): cell 0's record, loaded and never used.
        const int id_of_lane = kElectLeaders ? 0 : __builtin_amdgcn_ds_permute(need ? (slot << 2) : 252, nb);
        C5_STAMP(0);  // lanes -> slots
        // 2. cooperative loads into registers: kPasses passes of 8 slots (8 lanes x 16 B per record)
        //    + one pass for the optics (2 lanes per slot).  Measured on the C3 frame (same GPU): 24
        //    slots 1.31 ms, 32 slots 1.24 ms, 40 slots 1.38 ms and 48 slots 1.31 ms (registers); a
        //    second staging round instead of the direct-load fallback 1.41 ms.
        //    A pass none of whose slots is in use is skipped (wave-uniform branch).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wsometimes-uninitialized"
#pragma clang diagnostic ignored "-Wconditional-uninitialized"
        V2 stage_r[kPasses];  // a skipped pass leaves its register undefined; it is never stored either
#pragma clang diagnostic pop
        if (DMA) {
            const uint32_t ids_end = (uint32_t)(uintptr_t)(LdsInts)(my_elect + kBuckets1 + 64) + 4u * static_cast<uint32_t>(n_staged);
            uint32_t id_of_pass[kDmaPasses];
#pragma unroll
            for (int j = 0; j < kDmaPasses; ++j) id_of_pass[j] = static_cast<uint32_t>(((LdsInts)(uintptr_t)dma_id_at)[kDmaSlots * j]);
#pragma unroll
            for (int j = 0; j < kDmaPasses; ++j) {
                // (the lanes' own test below says it all; a wave-uniform "does the pass reach a staged slot at all" in front of
                // it cost three scalar instructions per step and saved a skipped branch in half of them)
                if (C5_LOOP_TRIM || j == 0 || kDmaSlots * j < n_staged) {
                    if (dma_id_at < ids_end - 4u * kDmaSlots * j) {  // this lane's slot kDmaSlots j + s is staged (idle lanes: never)
                        const uint32_t off = (id_of_pass[j] << 7) + dma_off;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rec_bytes + off),
                                                         (__attribute__((address_space(3))) void*)(my_stage + kDmaSlots * kSlotStride * j), 16, 0, 0);
                    }
                }
            }
        } else {
#pragma unroll
            for (int pass = 0; pass < kPasses; ++pass) {  // fully unrolled: stage_r[] stays in registers
                if (pass == 0 || 8 * pass < n_staged) {
                    const uint32_t id_ = static_cast<uint32_t>(
                        kElectLeaders ? my_elect[kBuckets1 + 64 + 8 * pass + sub] : __builtin_amdgcn_ds_bpermute(sub4 + 32 * pass, id_of_lane));
                    stage_r[pass] = *reinterpret_cast<const V2*>(rec_bytes + ((id_ << 7) | rec_piece_off));
                }
            }
        }
        C5_STAMP(1);  // bpermutes landed, five loads issued

        // ... while they are in flight: emission/absorption of the step just taken
        if (!kEmitNow) {
            // (wave-uniform choice of the exp: every pending lane's argument within (-1/8, 0] -> the short series)
            const bool big_arg = pend && !(pend_o0.y * pend_dz < -kSmallExpArg);
            const bool short_exp = SMALLEXP || __builtin_amdgcn_ballot_w64(big_arg) == 0ull;
            if (pend) {
                if (ORDER == 0) {
                    if (pend_o0.y != 0.0)  // line.cpp:220-224
                        I = short_exp ? reference_emission_step<true>(I, pend_o0.y, pend_o1.y, pend_o1.x, pend_dz)
                                      : reference_emission_step<false>(I, pend_o0.y, pend_o1.y, pend_o1.x, pend_dz);
                } else if (T >= P.t_cutoff) {
                    const double ex = short_exp ? exp_small_nonpositive(-pend_o0.y * pend_dz) : exp_nonpositive(-pend_o0.y * pend_dz);
                    I = fma(T * pend_o1.x, 1.0 - ex, I);
                    T *= ex;
                }
                pend = false;
            }
        }
        C5_STAMP(2);  // emission/absorption of the previous step

        // 3. park the pieces in LDS (the previous step's reads are long done: same wavefront, in order)
        __builtin_amdgcn_wave_barrier();
        if (DMA) {
            // nothing orders a ds_read behind this wavefront's own LDS-DMA but its vmcnt
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int pass = 0; pass < kPasses; ++pass)
                if (sub < n_staged - 8 * pass) put_rec[8 * pass * kSlotStride] = stage_r[pass];
        }
        __builtin_amdgcn_wave_barrier();
#if C5_WALK_STAMPS
        if (stamping_) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        C5_STAMP(3);  // loads landed, pieces parked in LDS
#elif defined(C5_ISA_MARKERS)
        C5_STAMP(3);
#endif

        // 4. every ray fetches its cell: eight 16-byte reads (the planes it can leave through, the neighbours behind
        //    them, the optics)
        if (need) {
            CellRegs cur;
            const V2* r = reinterpret_cast<const V2*>(reinterpret_cast<const char*>(my_stage) +
                                                      __umul24(static_cast<unsigned>(slot), kSlotStride * 16u));
            auto read_staged = [&]() {
                cur.r0 = as_d2(r[0]);
                cur.r1 = as_d2(r[1]);
                cur.r2 = as_d2(r[2]);
                cur.r3 = as_d2(r[3]);
                cur.r4 = as_d2(r[4]);
                cur.r5 = as_d2(r[5]);
                // the optics ride along, straight into the pending registers: those are free here (the
                // previous step's emission is done) and only looked at again if this step contributes
                pend_o0 = r[6];
                pend_o1 = r[7];
            };
            if (C5_LOOP_TRIM && n_runs_s <= kStageSlots) {
                // (wave-uniform, nine steps in ten: every distinct cell has a slot - no per-lane "is my cell staged" at all)
                read_staged();
            } else if (slot < kStageSlots) {
                read_staged();
            } else {
                load_cell(cur, P.xrec, nb);  // more distinct cells than slots: rare
                pend_o0 = V2{cur.r6.a, cur.r6.b};
                pend_o1 = V2{cur.r7.a, cur.r7.b};
            }

            const StepGeometry sg = step_geometry(cur, x, y);
            // SPLIT: a cell that reaches beyond the job's slab is cut at the plane: the job ends there, and the job above takes
            // the ray up from the plane (in this cell, if plane_raster found it there)
            const bool clip = SPLIT && sg.w_exit >= w_hi && sg.w_exit < INFINITY;
            const double dz = (clip ? w_hi : sg.w_exit) - carry;  // line.cpp:124-131: the chord through the cell
            const bool contributes = dz > 0.0 && dz < INFINITY;
            // (wave-uniform choice of the exp: every contributing lane's argument within (-1/8, 0] -> the short series;
            // C3 walk 0.4786 -> 0.4738 ms, at 4800x3600 1.614 -> 1.579.  Starting the next step's election — ticket
            // written, winner read — before this arithmetic, so that the LDS round trip runs under it, cost a
            // register pair to scratch and 0.7 %: not kept)
            // With SMALLEXP (the host knows that no cell of the grid can give an argument beyond -1/8) neither the test nor
            // the general exp is in the kernel: C3 walk 0.4735 -> 0.460 ms.  (The same knowledge as a per-frame flag
            // read by the kernel, both paths kept, bought nothing: 0.4747 against 0.4741.)
            const bool short_exp = SMALLEXP || !kEmitNow || __builtin_amdgcn_ballot_w64(contributes && !(pend_o0.y * dz < -kSmallExpArg)) == 0ull;
            if (contributes) {
                // (SPLIT: a cell is counted by the job in which the ray LEAVES it, so that a cell cut by a plane counts once)
                if (!clip) ++n_seg;
                tau = fma(dz, pend_o0.x, tau);  // line.cpp:189 (unclamped alpha); order-independent, done now
                if (SPLIT) tauc = fma(dz, pend_o0.y, tauc);
                if (kEmitNow) {
                    if (ORDER == 0) {
                        if (pend_o0.y != 0.0)  // line.cpp:220-224
                            I = short_exp ? reference_emission_step<true>(I, pend_o0.y, pend_o1.y, pend_o1.x, dz)
                                          : reference_emission_step<false>(I, pend_o0.y, pend_o1.y, pend_o1.x, dz);
                    } else if (T >= P.t_cutoff) {
                        const double ex = short_exp ? exp_small_nonpositive(-pend_o0.y * dz) : exp_nonpositive(-pend_o0.y * dz);
                        I = fma(T * pend_o1.x, 1.0 - ex, I);
                        T *= ex;
                    }
                } else {
                    pend = true;
                    pend_dz = dz;
                }
            }
            // the ray enters the next cell where it leaves this one (a cell it cannot leave — flat, or all its
            // candidates edge-on — has w_exit = +inf and no neighbour: the ray ends there, as before)
            const bool has_exit = sg.w_exit < INFINITY;
            if (has_exit) carry = sg.w_exit;
            const uint32_t id = sg.w_out & kIdMask;
            int nxt = static_cast<int>(id);
            if (id == kNoCell || clip) {  // left the grid: re-entry of a non-convex grid?  (or reached the end of the job's slab)
                const size_t lp = pixel_index_rare();
                double w_cur = my_scur[lane];  // (the key of the entry the ray took last: only (re-)entries write it)
                const double key_taken = w_cur;
                bool skipped = false;
                if (has_exit) w_cur = fmax(w_cur, sg.w_exit);
                // (a clipped ray takes no further entry - own_hi = -DBL_MAX owns none - but the entries inside the stretch it
                // has walked, up to the plane, are judged all the same)
                nxt = next_entry<kUp>(P, lp, load_entry_head(P.entry_head + lp), w_cur, carry, key_taken,
                                      clip ? w_hi : (has_exit ? sg.w_exit : -DBL_MAX), skipped, slab_lo(x, y), clip ? -DBL_MAX : w_hi);
                if (SPLIT && clip) n_seg |= kClippedBit;
                if (skipped) n_seg |= kSkippedBit;
                my_scur[lane] = w_cur;
            }
#if C5_WALK_STAMPS > 1
            nb_before_ = nb;
#endif
            nb = nxt;
        }
#if C5_WALK_STAMPS > 1
        if (stamping_) {  // how often is a ray's NEXT cell one this step already staged (some lane's current cell)?
            bool hit = false;
            for (int l = 0; l < 64; ++l) {
                const int v = __builtin_amdgcn_readlane(nb_before_, l);
                if (v >= 0 && need && nb == v) hit = true;
            }
            stat_next_staged += static_cast<unsigned>(__popcll(__builtin_amdgcn_ballot_w64(hit)));
            stat_next_total += static_cast<unsigned>(__popcll(__builtin_amdgcn_ballot_w64(need && nb >= 0)));
        }
#endif
#if C5_WALK_STAMPS
        if (stamping_) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        C5_STAMP(4);  // record read back, geometry, exit face, (re-entry)
#elif defined(C5_ISA_MARKERS)
        C5_STAMP(4);
#endif
    }
#if C5_WALK_STAMPS
    if (lane == 0 && blockIdx.x < 131072u) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned long long* t = g_walk_trace + 4ull * blockIdx.x;
        t[0] = trace_begin_;
        t[1] = __builtin_amdgcn_s_memrealtime();
        t[2] = (static_cast<unsigned long long>(hw) << 8) | (xcc & 0xffu);
        t[3] = stat_iters;
    }
    if (lane == 0 && stamping_) {
        for (int k = 0; k < 5; ++k) atomicAdd(&g_walk_stamps[k], stamp_acc[k]);
        atomicAdd(&g_walk_stamps[8], __builtin_amdgcn_s_memtime() - t_begin_);  // whole loop
        atomicAdd(&g_walk_stamps[9], 1ull);                                      // wavefronts
        atomicAdd(&g_walk_stamps[10], static_cast<unsigned long long>(stat_runs));
        atomicAdd(&g_walk_stamps[11], static_cast<unsigned long long>(stat_distinct));
#if C5_WALK_STAMPS > 1
        atomicAdd(&g_walk_stamps[14], static_cast<unsigned long long>(stat_next_staged));
        atomicAdd(&g_walk_stamps[15], static_cast<unsigned long long>(stat_next_total));
#endif
        atomicAdd(&g_walk_stamps[12], static_cast<unsigned long long>(stat_iters));
        atomicAdd(&g_walk_stamps[13], static_cast<unsigned long long>(stat_lanes));
    }
#endif
    if (pend) {  // the last step's contribution
        if (ORDER == 0) {
            if (pend_o0.y != 0.0) I = reference_emission_step<SMALLEXP>(I, pend_o0.y, pend_o1.y, pend_o1.x, pend_dz);
        } else if (T >= P.t_cutoff) {
            const double ex = exp_nonpositive(-pend_o0.y * pend_dz);
            I = fma(T * pend_o1.x, 1.0 - ex, I);
            T *= ex;
        }
    }

    {   // the depth sample (DepthSamples): where the rays of the sampled pixels END - in this job, if it did not hand them on
        const bool ended_here = (n_seg & ~(kOverflowBit | kSkippedBit | kClippedBit)) != 0u && (n_seg & kClippedBit) == 0u;
        int l = lane;
        asm volatile("" : "+v"(l));
        const int col = tx * TW + (wave % TS::GX) * TS::WW + (l % TS::WW), lrow = ty * TH + (wave / TS::GX) * TS::WH + (l / TS::WW);
        if (ended_here && col < im.res_x && lrow < im.n_local_rows) {
            const int slot = fit_slot_of(im, col, global_row_of(im, lrow));
            if (slot >= 0) reinterpret_cast<DepthSamples*>(P.counters + kCounterShards)->exit_key[slot] = depth_key(carry);
        }
        n_seg &= ~kClippedBit;
    }
    {   // ... and the deepest at which one of them ends (carry: where the ray left its last cell)
        double hi = (n_seg & ~(kOverflowBit | kSkippedBit)) ? carry : -DBL_MAX;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) hi = fmax(hi, __shfl_xor(hi, d));
        if (lane == 0 && hi > -DBL_MAX && sampler) atomicMax(&fc_job()->exit_max_key, depth_key(hi));
    }
    if (SPLIT) {
        // The job's partial results, indexed tile * 64 + lane (whole 512-byte rows per array and wavefront), then the
        // tile's arrival count: the job that finds K - 1 others already there composes the K partials in depth order and
        // goes on to the common end (image, entry heads, statistics).  The partials travel as RELAXED ATOMIC stores and
        // loads at agent scope (write-through / read-through: coherent across the XCDs' L2s by themselves) with the
        // wavefront's own vmcnt between its stores and its arrival — NOT as plain stores behind a release fence: an
        // agent-scope release writes back the WHOLE L2 (buffer_wbl2), and 80 000 of those per launch made the first
        // version of this kernel five times slower than the walk it replaces (profiles/r04_split_probe.md).
        const unsigned K = static_cast<unsigned>(P.split.n_slabs);
        const size_t tile = static_cast<size_t>(ty) * tiles_x + tx;
        const size_t at = tile * 64u + static_cast<unsigned>(lane);
        const size_t mine = static_cast<size_t>(slab) * P.split.part_stride + at;
        __hip_atomic_store(P.split.part_tau + mine, tau, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(P.split.part_tauc + mine, tauc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(P.split.part_b + mine, I, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(P.split.part_nseg + mine, n_seg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the four stores above have been performed
        unsigned before = 0;
        if (lane == 0) before = atomicAdd(P.split.arrivals + tile, 1u);
        before = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(before)));
        if (before + 1u != K) {
            if (lane == 0 && n_step_wave)
                atomicAdd(&(P.counters + ((blockIdx.x * 4u + static_cast<unsigned>(wave)) % kCounterShards))->steps_cov,
                          static_cast<unsigned long long>(n_step_wave));
            return;
        }
        if (lane == 0) P.split.arrivals[tile] = 0u;  // (for the next frame; every job of this one has arrived)
        tau = 0.0;
        I = 0.0;
        n_seg = 0u;
        for (unsigned k = 0; k < K; ++k) {  // line.cpp:206-225 is affine in I: I <- exp(-tauc_k) I + b_k, from the back
            const size_t from = static_cast<size_t>(k) * P.split.part_stride + at;
            const double t_k = __hip_atomic_load(P.split.part_tau + from, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double c_k = __hip_atomic_load(P.split.part_tauc + from, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double b_k = __hip_atomic_load(P.split.part_b + from, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned n_k = __hip_atomic_load(P.split.part_nseg + from, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tau += t_k;
            I = fma(exp_nonpositive(-c_k), I, b_k);
            n_seg = (n_seg + (n_k & ~(kOverflowBit | kSkippedBit))) | (n_k & (kOverflowBit | kSkippedBit));
        }
    }
    const unsigned overflow = n_seg >> 31;
    const bool skipped = (n_seg & kSkippedBit) != 0u;
    n_seg &= ~(kOverflowBit | kSkippedBit);
    unsigned is_solid = 0, n_entries = 0;
    if (in_image) {
        const size_t lp = pixel_index();
        float2 result = make_float2(static_cast<float>(tau), static_cast<float>(I));  // plane.cpp:165-166
        const uint32_t mv = P.mask ? P.mask[lp] : 0u;
        if (mv) {
            double colour = 0.0;
            for (int s = 0; s < P.solids.n_slots; ++s)
                if (mv == static_cast<uint32_t>(s) + 1u) colour = P.solids.colour[s];
            result.x = static_cast<float>(colour);
            result.y = result.x;
            is_solid = 1;
        }
        // every pixel hands its entry head back cleared: the next frame's raster needs no memset
        n_entries = static_cast<unsigned>(load_entry_head(P.entry_head + lp).count);
        if (n_entries && !P.keep_entries) __builtin_nontemporal_store(0ll, reinterpret_cast<long long*>(P.entry_head + lp));
        __builtin_nontemporal_store(result.x, &P.out[lp].x);
        __builtin_nontemporal_store(result.y, &P.out[lp].y);
    }

    if (P.row_cost) {
        int l = lane;
        asm volatile("" : "+v"(l));  // (as in pixel_index_rare: nothing of this is kept across the loop)
        const int lrow = ty * TH + (wave / TS::GX) * TS::WH + (l / TS::WW);
        unsigned rs = n_seg;
#pragma unroll
        for (int d = TS::WW / 2; d >= 1; d >>= 1) rs += __shfl_xor(rs, d);
        if ((l % TS::WW) == 0 && rs && lrow < im.n_local_rows) atomicAdd(P.row_cost + lrow, rs);
    }
    const unsigned s_seg = wave_sum_u32(n_seg);
    const unsigned s_cov = wave_sum_u32(n_seg > 0 ? 1u : 0u);
    const unsigned s_sol = wave_sum_u32(is_solid);
    const unsigned s_ovf = wave_sum_u32(overflow);
    const unsigned s_ent = wave_sum_u32(n_entries);
    const unsigned s_skip = static_cast<unsigned>(__popcll(__builtin_amdgcn_ballot_w64(skipped)));
    unsigned s_longest = n_seg;  // the longest ray of the tile, in segments
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s_longest = max(s_longest, static_cast<unsigned>(__shfl_xor(static_cast<int>(s_longest), d)));
    if (lane == 0) {
        if (s_skip) {
            atomicAdd(&P.counters->overlap_rays, s_skip);
            atomicAdd(P.sticky + 2, s_skip);
        }
        FrameCounters* const fc = fc_job();
        // The longest ray of this tile, to its row of super-blocks (the order the next frames' rows start in, c_api.hip) - from ONE
        // tile per super-block only (its first: the one a cut-off super-block at the image's edge has too): every wavefront
        // adding to its row's word put thousands of same-address atomics on a handful of words (a share of 248 rows has 8
        // rows of super-blocks: walk 0.219 -> 0.231 ms; the C3 frame + 0.7 %).
        if (s_seg) atomicMax(&fc->seg_max, s_longest);  // (exact: "depth_split" 0 decides by it, and the decision has a threshold)
        if (s_seg && sampler) {
            const int sb_row = ty / P.band_tiles;
            if (P.sb_cost && P.xcd_mode == 2 && sb_row < kMaxSbRows) atomicMax(P.sb_cost + sb_row, s_longest);
        }
        if (s_seg) atomicAdd(&fc->seg_tiles, static_cast<unsigned long long>(s_seg) | (1ull << kCounterHighShift));
        if (n_step_wave | s_cov) atomicAdd(&fc->steps_cov, static_cast<unsigned long long>(n_step_wave) | (static_cast<unsigned long long>(s_cov) << kCounterHighShift));
        if (s_ent | s_sol) atomicAdd(&fc->ent_solid, static_cast<unsigned long long>(s_ent) | (static_cast<unsigned long long>(s_sol) << kCounterHighShift));
        if (s_ovf) {  // (shard 0, beside entry_overflow: the two words a frame's status is read from; rare, never contended)
            atomicAdd(&P.counters->walk_overflow, s_ovf);
            atomicAdd(P.sticky + 1, s_ovf);
        }
    }
}

#if C5_WALK_STAMPS
extern "C" int c5_debug_walk_trace(unsigned long long* out, int n_blocks, int reset) {
    if (n_blocks > 131072) n_blocks = 131072;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_walk_trace), 32ull * n_blocks) != hipSuccess) return 1;
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_walk_trace)) != hipSuccess) return 1;
        if (hipMemset(p, 0, sizeof(unsigned long long) * 4 * 131072) != hipSuccess) return 1;
    }
    return 0;
}
extern "C" int c5_debug_walk_stamps(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_walk_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_walk_stamps), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

// the instantiation without the general exp (SMALLEXP), kept to the default tile shape: true if it took the launch
template <int TILE, int ORDER>
static bool launch_small_exp(hipStream_t s, const WalkParams& q, long long blocks, unsigned threads) {
    if constexpr (TILE == 3) {
        if (q.lds_stage == 2 && q.small_exp_only && q.stage_slots <= 14) {
            hipLaunchKernelGGL((walk_composite_lds<3, ORDER, true, kStageSlots, true>), dim3(static_cast<unsigned>(blocks)), dim3(threads),
                               static_cast<size_t>(q.lds_pad), s, q);
            return true;
        }
    }
    return false;
}

template <int TILE, int ORDER>
static void launch_walk_t(hipStream_t s, const WalkParams& p) {
    using TS = TileShape<TILE>;
    constexpr int TW = TS::WW * TS::GX, TH = TS::WH * TS::GY;
    constexpr unsigned kThreads = 64u * TS::GX * TS::GY;
    const int tiles_x = (p.im.res_x + TW - 1) / TW;
    const int tiles_y = (p.im.n_local_rows + TH - 1) / TH;
    if (tiles_x <= 0 || tiles_y <= 0) return;
    long long blocks;
    if (p.xcd_mode == 0) {
        blocks = static_cast<long long>(tiles_x) * tiles_y;
    } else if (p.xcd_mode == 2 && p.lds_stage) {
        // measured (C3 grid, 8x8 tiles, walk ms at 1200x900 / 2400x1800 / 4800x3600): super-blocks of 32 rows
        // 0.303 / 0.601 / 2.070, of 64 rows 0.350 / 0.602 / 2.063; bands of 16 rows 0.310 / 0.610 / 2.065;
        // row-major blocks 0.372 / 0.597 / 2.059
        const int sb_rows = p.band_rows > 0 ? p.band_rows : 32;
        const int S = sb_rows / TH > 0 ? sb_rows / TH : 1;
        const long long n_sb = static_cast<long long>((tiles_x + S - 1) / S) * ((tiles_y + S - 1) / S);
        blocks = 8ll * ((n_sb + 7) / 8) * S * S;
        WalkParams q = p;
        q.band_tiles = S;
        if constexpr (TILE == 3 && ORDER == 0) {
            if (p.split.n_slabs > 1 && p.lds_stage == 2) {  // "depth_split": K jobs per tile (the host only asks for it in this configuration)
                const unsigned grid = static_cast<unsigned>(blocks * p.split.n_slabs);
                if (p.stage_slots > 14)
                    hipLaunchKernelGGL((walk_composite_lds<3, 0, true, 21, false, true>), dim3(grid), dim3(kThreads), static_cast<size_t>(p.lds_pad), s, q);
                else if (p.small_exp_only)
                    hipLaunchKernelGGL((walk_composite_lds<3, 0, true, kStageSlots, true, true>), dim3(grid), dim3(kThreads), static_cast<size_t>(p.lds_pad), s, q);
                else
                    hipLaunchKernelGGL((walk_composite_lds<3, 0, true, kStageSlots, false, true>), dim3(grid), dim3(kThreads), static_cast<size_t>(p.lds_pad), s, q);
                return;
            }
        }
        if (p.lds_stage == 2 && p.stage_slots > 14)
            hipLaunchKernelGGL((walk_composite_lds<TILE, ORDER, true, 21>), dim3(static_cast<unsigned>(blocks)), dim3(kThreads), static_cast<size_t>(p.lds_pad), s, q);
        else if (launch_small_exp<TILE, ORDER>(s, q, blocks, kThreads))
            ;
        else if (p.lds_stage == 2)
            hipLaunchKernelGGL((walk_composite_lds<TILE, ORDER, true>), dim3(static_cast<unsigned>(blocks)), dim3(kThreads), static_cast<size_t>(p.lds_pad), s, q);
        else
            hipLaunchKernelGGL((walk_composite_lds<TILE, ORDER>), dim3(static_cast<unsigned>(blocks)), dim3(kThreads), static_cast<size_t>(p.lds_pad), s, q);
        return;
    } else {
        // ~16 image rows per band (one row of workgroups), but never fewer than 16 bands (2 per XCD) on a short strip
        // measured (C3 grid, 8x8 tiles, walk ms at 1200x900 / 2400x1800 / 4800x3600): row-major blocks
        // 0.378 / 0.629 / 2.140, bands of 16 rows 0.319 / 0.645 / 2.137, 32 rows 0.322 / 0.657 / 2.146
        const int band_rows = p.band_rows > 0 ? p.band_rows : 16;
        int band = (band_rows / TH) > 0 ? (band_rows / TH) : 1;
        while (band > 1 && (tiles_y + band - 1) / band < 16) band >>= 1;
        const int n_bands = (tiles_y + band - 1) / band;
        const int rounds = (n_bands + 7) / 8;
        blocks = 8ll * rounds * band * tiles_x;
        WalkParams q = p;
        q.band_tiles = band;
        if (p.lds_stage == 2 && p.stage_slots > 14)
            hipLaunchKernelGGL((walk_composite_lds<TILE, ORDER, true, 21>), dim3(static_cast<unsigned>(blocks)), dim3(kThreads), static_cast<size_t>(p.lds_pad), s, q);
        else if (launch_small_exp<TILE, ORDER>(s, q, blocks, kThreads))
            ;
        else if (p.lds_stage == 2)
            hipLaunchKernelGGL((walk_composite_lds<TILE, ORDER, true>), dim3(static_cast<unsigned>(blocks)), dim3(kThreads), static_cast<size_t>(p.lds_pad), s, q);
        else if (p.lds_stage)
            hipLaunchKernelGGL((walk_composite_lds<TILE, ORDER>), dim3(static_cast<unsigned>(blocks)), dim3(kThreads), static_cast<size_t>(p.lds_pad), s, q);
        else
            hipLaunchKernelGGL((walk_composite<TILE, ORDER>), dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, s, q);
        return;
    }
    if (p.lds_stage == 2 && p.stage_slots > 14)
        hipLaunchKernelGGL((walk_composite_lds<TILE, ORDER, true, 21>), dim3(static_cast<unsigned>(blocks)), dim3(kThreads), static_cast<size_t>(p.lds_pad), s, p);
    else if (p.lds_stage == 2)
        hipLaunchKernelGGL((walk_composite_lds<TILE, ORDER, true>), dim3(static_cast<unsigned>(blocks)), dim3(kThreads), static_cast<size_t>(p.lds_pad), s, p);
    else if (p.lds_stage)
        hipLaunchKernelGGL((walk_composite_lds<TILE, ORDER>), dim3(static_cast<unsigned>(blocks)), dim3(kThreads), static_cast<size_t>(p.lds_pad), s, p);
    else
        hipLaunchKernelGGL((walk_composite<TILE, ORDER>), dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, s, p);
}

int walk_sb_rows(int tile_shape, int band_rows) {
    const int th = tile_shape == 0 ? TileShape<0>::WH * TileShape<0>::GY
                   : tile_shape == 1 ? TileShape<1>::WH * TileShape<1>::GY
                   : tile_shape == 2 ? TileShape<2>::WH * TileShape<2>::GY
                                     : TileShape<3>::WH * TileShape<3>::GY;
    const int sb_rows = band_rows > 0 ? band_rows : 32;
    const int S = sb_rows / th > 0 ? sb_rows / th : 1;
    return S * th;
}

void launch_walk(hipStream_t s, const WalkParams& p, int tile_shape) {
    if (p.order == 0) {
        switch (tile_shape) {
            case 1: launch_walk_t<1, 0>(s, p); break;
            case 2: launch_walk_t<2, 0>(s, p); break;
            case 3: launch_walk_t<3, 0>(s, p); break;
            default: launch_walk_t<0, 0>(s, p); break;
        }
    } else {
        switch (tile_shape) {
            case 1: launch_walk_t<1, 1>(s, p); break;
            case 2: launch_walk_t<2, 1>(s, p); break;
            case 3: launch_walk_t<3, 1>(s, p); break;
            default: launch_walk_t<0, 1>(s, p); break;
        }
    }
}

}  // namespace c5
