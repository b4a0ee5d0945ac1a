// Device-side data layout of the MI355X render path (shared by host and kernels).
//
// Persistent grid (uploaded once, SoA, coalesced for the per-frame setup kernels):
//   px/py/pz[n_pts]      fp64   raw vertex coordinates
//   cell_vert[n_cells]   int4   cell -> vertex ids          (tetra.hpp:42 replaced by indices)
//   cell_adj[n_cells]    int4   cell -> neighbour across face f; boundary: -(i + 2), i = the face's index in bface[] (host API: -1)
//   alpha/q[n_cells]     fp64   AbsorpCoef / radEnLooseRate (object3d_accretion_disk.cpp:4)
//   bface[n_bfaces]      u32    (cell << 2 | face) of every face without a neighbour
//
// Per view (rebuilt every frame by build_records from the transformed vertices):
//   CellRecord[n_cells]  128 B  one cache line: everything one walk step needs geometrically
//   CellOptics[n_cells]   32 B  clamped absorption + source function
//
// Face numbering and vertex order follow the reference (plane.cpp:16-21,30-37; line.cpp:103-122):
//   face 0 = (0,1,2), face 1 = (0,1,3), face 2 = (0,2,3), face 3 = (1,2,3).
#pragma once

#include <cstdint>

namespace c5 {

constexpr uint32_t kIdMask = 0x0FFFFFFFu;   // 28-bit cell id (line.hpp:71-79, line.cpp:27)
constexpr uint32_t kNoCell = 0x0FFFFFFFu;   // neighbour field of a boundary face
constexpr int kUpperCountShift = 30;        // nbr[0] bits 30-31: number of "upper" slots (1..3)

// One walk step needs, per face, z(x, y) = plane[k][0] + plane[k][1] * (x - x0) + plane[k][2] * (y - y0)
// (line::find_polygon_intersection_z, line.cpp:150-174, rewritten about the cell-local origin
// (x0, y0) = vertex 0 so that the 128-byte record holds all four faces) and the neighbour behind it.
//
// The four faces are stored in WALK ORDER, not in the reference's face numbering: first the
// n_up "upper" faces (the cell body lies below their plane), then the "lower" ones, so the kernel
// classifies a slot by its position.  z_top = min over upper slots, z_bot = max over lower slots are
// the two faces the reference pairs for a pixel inside the cell's projection (line.cpp:99-131).
// Faces that are edge-on to the rays (no z(x, y)) are stored as upper slots with plane +inf so they
// never win; a flat cell gets +inf / -inf in slots 0 / 3 and therefore neither contributes nor
// forwards the ray.
struct alignas(16) CellRecord {
    double x0, y0;
    double plane[4][3];
    uint32_t nbr[4];  // kIdMask bits: neighbour cell or kNoCell; nbr[0] also carries n_up
};
static_assert(sizeof(CellRecord) == 128, "CellRecord must be one 128-byte line");

// line.cpp:204-224 folded per cell: alpha_c = min(alpha, limit); cells with alpha_c < DBL_EPSILON
// neither absorb nor emit (alpha_c = 0).  `aux` is what the chosen integration order multiplies by:
// 1 / alpha_c for the reference's recurrence (its final division, as a reciprocal: <= 1 ulp apart),
// Q / alpha_c (the source function) for the front-to-back sum.
struct alignas(16) CellOptics {
    double alpha_raw;  // ch0 uses the unclamped value (line.cpp:189)
    double alpha_c;
    double aux;
    double q;
};
static_assert(sizeof(CellOptics) == 32, "CellOptics is 32 bytes");

// What the fp64 walk reads per step since round 3: ONE 128-byte line per cell and view with only what a ray needs to
// LEAVE the cell, and the cell's optics beside it.
//   * exit candidates only: walking along +z (reference order: back to front) a ray leaves through the lowest of the
//     cell's upper faces, walking along -z through the highest of its lower ones; a tetrahedron has at most three of
//     either.  The depth at which the ray ENTERED is the depth at which it left the cell before (or the boundary
//     entry's own depth): the chord is the difference of two consecutive exit depths, so the entry-side planes are
//     not evaluated at all — 3 planes instead of 4, no upper / lower classification of slots.
//   * walk coordinate w (grows along the walk): w = z for order 0, w = -z for order 1 (the candidates of order 1 are
//     stored negated), so that both orders are "w_exit = min over the candidates".
//   * planes about the ABSOLUTE pixel coordinates, w(x, y) = c + gx x + gy y: no cell-local origin in the record and
//     no (x - x0, y - y0) per step.  Against the cell-local form this costs rounding of the size eps * kappa * |x|
//     (kappa = the face's slope against the rays) instead of eps * kappa * extent — far below the fp32 output either
//     way, and what the entry keys' slack for steep faces already allows for.
//   * unused candidates: c = +inf, gx = gy = 0 (never the minimum); edge-on faces and flat cells likewise.
// 16-byte units: 0-4 planes (+ nbr[0..1] in the upper half of unit 4), 5 = {nbr[2], flags, pad}, 6 = {alpha_raw,
// alpha_c}, 7 = {aux, q} (CellOptics).
struct alignas(16) ExitRecord {
    double plane[3][3];
    uint32_t nbr[3];  // kIdMask bits: neighbour behind candidate k, or kNoCell
    uint32_t flags;   // spare
    double pad;
    double alpha_raw, alpha_c, aux, q;
};
static_assert(sizeof(ExitRecord) == 128, "ExitRecord must be one 128-byte line");

// A place where a ray enters the grid through a boundary face.  first[pixel] holds a pixel's first
// entry; further ones live in the overflow pool, chained through `next` (pool slot + 1, 0 = end).
// z: the entry face's own depth at the pixel.  cell: bits 0-27 the cell, bits 28-31 k: the entry is KEYED at its
// depth pushed (frame's uniform slack) * 2^k along the walk (k > 0 only for faces steep against the rays;
// walk_common.hpp: next_entry, entry_key_slack).
struct alignas(16) Entry {
    double z;
    uint32_t cell;
    int32_t next;
};
constexpr int kEntrySlackShift = 28;
// What entry_raster_rec needs of one boundary face, left by build_records for the faces a ray can ENTER through under
// this view (the cell's record is being built anyway: vertices and face planes are in registers there): the face's three
// projected vertices, its plane z = pc + pgx (x - x0) + pgy (y - y0), the cell word of its entries (id + key exponent) and
// the number of the frame that wrote it - a record of another frame (a face turned away, a cell outside this context's
// rows) is not rastered.  Slot i belongs to the i-th face of the sorted boundary-face list (GridView::bface); the
// device's adjacency table holds -(i + 2) where the host's holds -1.
struct alignas(16) BFaceRecord {
    double ax, ay, bx, by, cx, cy, x0, y0, pc, pgx, pgy;
    uint32_t cell_word;
    uint32_t seq;
};
static_assert(sizeof(BFaceRecord) == 96, "BFaceRecord is six 16-byte units");
// Per pixel and frame: number of entries and the head of the pixel's overflow chain (pool slot + 1).
// Cleared to zero before every raster pass.
struct alignas(8) EntryHead {
    int32_t count;
    int32_t chain;
};

// --- "depth_split" (round 4): a ray cut at planes of constant depth ------------------------------------------------
// A ray is a chain of ~170 dependent steps of ~1.1 us; a frame (or one GPU's share of a frame) with fewer rays than the
// GPU has wavefront slots lasts as long as that chain whatever its size.  With n_slabs = K > 1 the depth range of the
// grid is cut at K - 1 planes w[1] < ... < w[K - 1] of the walk coordinate and every 8x8 pixel tile becomes K jobs: job s
// walks its rays from w[s] to w[s + 1] (job 0 from the boundary entries, the last one until the ray leaves the grid for
// good) and leaves per pixel a PARTIAL result - tau_s, the clamped optical depth tauc_s, and b_s = the recurrence of
// line.cpp:206-225 started from I = 0 - and the tile's last job to arrive composes them in depth order:
//     tau = sum tau_s,   I <- exp(-tauc_s) * I + b_s   (s = 0 ... K - 1; the recurrence is affine in I).
// Where a ray is at depth w[s] is found by plane_raster: the cells that straddle the plane (listed by build_records)
// are scan-converted like boundary faces, plane_cell[s - 1][pixel] = stamp << 28 | cell.
constexpr int kMaxSlabs = 8;
constexpr int kStraddleShards = 64;
constexpr int kStraddleCounterStride = 32;  // u32 words: one 128-byte line per counter
struct SplitParams {
    int32_t n_slabs;            // K; 0 / 1: rays are walked whole
    uint32_t stamp;             // 1..15: a plane_cell word is valid iff its bits 28-31 hold this (no clearing per frame)
    double w[kMaxSlabs + 1];    // w[0] = -DBL_MAX, w[K] = +DBL_MAX: the planes' depths at x = y = 0 ...
    double gx, gy;              // ... and their common tilt: plane s at pixel (x, y) = w[s] + gx x + gy y (0, 0: planes of constant depth)
    uint32_t* plane_cell;       // [K - 1][plane_stride]
    int64_t plane_stride;       // pixels of the local image, padded
    uint32_t* straddle;         // kStraddleShards lists of (cell | (plane - 1) << 28), appended to by build_records
    uint32_t* straddle_count;   // [shard * kStraddleCounterStride] items in the shard's list: the half build_records fills this frame
    uint32_t* straddle_count_next;  // the other half: zeroed by this frame's plane_raster for the next frame
    uint32_t straddle_capacity; // per shard: 64 (K - 1) x the wavefronts of build_records that belong to it
    double* part_tau;           // [K][part_stride] partial results, indexed tile * 64 + lane
    double* part_tauc;
    double* part_b;
    uint32_t* part_nseg;        // segments counted by the job (+ flag bits 30, 31)
    uint32_t* arrivals;         // [tiles] jobs of the tile that have delivered their partials (the last one composes, and zeroes it)
    int64_t part_stride;        // tiles * 64
};

constexpr int kMaxRotations = 8;
struct RotationList {
    int32_t n;
    int32_t axis[kMaxRotations];
    double cosv[kMaxRotations];
    double sinv[kMaxRotations];
    double x0[kMaxRotations];
};

struct ImageParams {
    int32_t res_x, res_y;     // full image
    int32_t n_local_rows;     // rows rendered by this context
    int32_t tile_rows, rank, world;  // row tile t (counted from row_begin) belongs to rank t % world
    int32_t row_begin, row_count;    // only rows [row_begin, row_begin + row_count) are rendered at all
    int32_t fit_shift, fit_cols;     // the depth sample (DepthSamples): one pixel per 2^shift x 2^shift box, fit_cols boxes per row
    double x_min, y_min;      // bounds[1], bounds[3]
    double step_x, step_y;    // plane.cpp:298-302
};

constexpr int kMaxSolids = 8;
struct SolidTable {
    int32_t n_slots;
    double colour[kMaxSolids];  // mask value v > 0 means "covered by solid slot v - 1"
};

// Per-frame statistics.  The device holds kCounterShards copies, each on a 128-byte line of its own:
// thousands of wavefronts adding to ONE address serialise at ~10 ns per atomic (three per covered
// wavefront used to put a 0.7 ms floor under the C3 walk); spread over 64 lines they vanish.  The host
// sums the shards (finish_frame).  Single-instance fields (odd_pixels, the overflow words) live in
// shard 0; pool_used is the allocation counter of shard k's part of the overflow pool (entry_raster).
constexpr int kCounterShards = 64;
// The sums travel two to a 64-bit word (one atomic per pair): a wavefront of the walk ends on three adds instead of seven
// - seven cost the C3 walk 2.2 % (0.530 -> 0.518 ms without them), the wavefront's slot is not free before they are out.
// Low field: 40 bits per shard (1.1e12 segments), high field 24 bits per shard (16.7 M pixels, i.e. images of up to 1 Gpx).
constexpr int kCounterHighShift = 40;
constexpr unsigned long long kCounterLowMask = (1ull << kCounterHighShift) - 1ull;
struct alignas(128) FrameCounters {
    unsigned long long seg_tiles;  // ray-cell segments | wavefront tiles with at least one segment ("depth_split" 0 looks at it) << 40
    unsigned long long steps_cov;  // lane-steps | covered pixels << 40
    unsigned long long ent_solid;  // boundary entries of the pixels rendered | pixels covered by a solid << 40
    // shard 0 only, three adjacent words = a frame's status (c_api.hip reads them behind a frame delivered to host memory):
    unsigned int walk_overflow;   // rays that hit the step bound
    unsigned int entry_overflow;  // boundary entries that found no slot in the overflow pool this frame
    unsigned int overlap_rays;    // rays that had to SKIP a boundary entry lying inside a stretch they had walked: cells of two
                                  // components interpenetrate there (walk_common.hpp: next_entry) - not a grid a walk can render
    unsigned int odd_pixels;  // bin_sort_resolve: (pixel, cell) pairs with an odd number of covering faces
    unsigned int pool_used;   // per shard: slots asked of this shard's part of the overflow pool (may exceed the part)
    unsigned int seg_max;     // most segments of any ray (per job of a cut ray: of any part)
    // from ONE tile in sixteen (the first of every 4 x 4 super-block; what they feed are estimates):
    // the depths between which this frame's rays ran, as keys that atomicMax orders (0: no ray): depth_key() of the deepest
    // exit, and of the NEGATED shallowest entry ("depth_split" 0 places the next frame's cutting planes between them)
    unsigned long long exit_max_key;
    unsigned long long entry_min_key;
};
// Behind the kCounterShards FrameCounters, 32 more 128-byte lines (cleared and copied to the host with them): a SAMPLE of
// the frame's rays - the middle pixel of every 2^fit_shift x 2^fit_shift box of the full image, at most kFitSlots of them,
// each with a slot of its own (plain stores: sums by atomicAdd serialise on their few addresses, 0.4 ms at 4800x3600) -
// with the depth at which the ray enters the grid (entry_raster) and the depth at which it ends (walk), as depth_key()s
// (0: none).  The host fits a plane through either set: "depth_split" 0 tilts the next frame's cutting planes with them, so
// that an oblique view's rays are cut at equal fractions (c_api.hip, finish_frame).
constexpr int kFitSlots = 256;
struct alignas(128) DepthSamples {
    unsigned long long entry_key[kFitSlots];
    unsigned long long exit_key[kFitSlots];
};
constexpr int kCounterLines = kCounterShards + static_cast<int>(sizeof(DepthSamples) / 128);

// monotone map double -> u64, never 0 for a finite value (0 = "none" after the per-frame clear)
__host__ __device__ inline unsigned long long depth_key(double w) {
    union { double d; unsigned long long u; } v;
    v.d = w;
    return (v.u >> 63) ? ~v.u : (v.u | 0x8000000000000000ull);
}
__host__ __device__ inline double depth_of_key(unsigned long long k) {
    union { double d; unsigned long long u; } v;
    v.u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return v.d;
}

// global row -> local row of this rank, or -1
__host__ __device__ inline int local_row_of(const ImageParams& im, int row) {
    const int r = row - im.row_begin;
    if (r < 0 || r >= im.row_count) return -1;
    if (im.world == 1) return r;  // no integer divisions on the common path
    const int tile = r / im.tile_rows;
    if (tile % im.world != im.rank) return -1;
    return (tile / im.world) * im.tile_rows + (r - tile * im.tile_rows);
}
// Local rows whose global row lies in [g0, g1] (both inside [row_begin, row_begin + row_count)) form one
// range of consecutive LOCAL indices, because local rows are numbered in global order: its first and
// last index, first > last if there is none.  Lets the entry raster enumerate only this context's rows.
__host__ __device__ inline void local_row_span(const ImageParams& im, int g0, int g1, int& first, int& last) {
    const int a = g0 - im.row_begin, b = g1 - im.row_begin;
    if (im.world == 1) {
        first = a;
        last = b;
        return;
    }
    // first local row at or after a
    const int ta = a / im.tile_rows;
    const int skip_a = ((im.rank - ta % im.world) + im.world) % im.world;  // tiles to the next one of this rank
    first = (skip_a == 0) ? (ta / im.world) * im.tile_rows + (a - ta * im.tile_rows)
                          : ((ta + skip_a) / im.world) * im.tile_rows;
    // last local row at or before b
    const int tb = b / im.tile_rows;
    const int back_b = ((tb % im.world - im.rank) + im.world) % im.world;  // tiles back to the previous one of this rank
    if (back_b == 0) {
        last = (tb / im.world) * im.tile_rows + (b - tb * im.tile_rows);
    } else if (tb - back_b < 0) {
        last = -1;
    } else {
        last = ((tb - back_b) / im.world) * im.tile_rows + im.tile_rows - 1;
    }
    if (last > im.n_local_rows - 1) last = im.n_local_rows - 1;
}
// local row -> global row
__host__ __device__ inline int global_row_of(const ImageParams& im, int lrow) {
    if (im.world == 1) return im.row_begin + lrow;
    const int ltile = lrow / im.tile_rows;
    return im.row_begin + (ltile * im.world + im.rank) * im.tile_rows + (lrow - ltile * im.tile_rows);
}
// the slot of the depth sample (DepthSamples) a pixel fills, or -1
__host__ __device__ inline int fit_slot_of(const ImageParams& im, int col, int global_row) {
    const int mask = (1 << im.fit_shift) - 1, mid = mask >> 1;
    if ((col & mask) != mid || (global_row & mask) != mid || im.fit_cols <= 0) return -1;
    return (global_row >> im.fit_shift) * im.fit_cols + (col >> im.fit_shift);
}

}  // namespace c5
