// Host-callable launchers of the gfx950 kernels (definitions in exact_kernels.hip and
// walk_kernels.hip).  All launches are asynchronous on the given stream.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "device_types.hpp"

namespace c5 {

struct GridView {  // device pointers of the persistent grid + per-view buffers
    int64_t n_pts = 0, n_cells = 0, n_bfaces = 0;
    const double *px = nullptr, *py = nullptr, *pz = nullptr;
    double *vx = nullptr, *vy = nullptr, *vz = nullptr;
    const int4* cell_vert = nullptr;
    const int4* cell_adj = nullptr;
    const double *alpha = nullptr, *q = nullptr;
    const uint32_t* bface = nullptr;
    ExitRecord* xrec = nullptr;  // fp64 walk: one 128-byte line per cell and view (exit candidates + optics)
    // only cells whose projected y-extent meets [cull_y_lo, cull_y_hi] can be reached by a ray of this context
    double cull_y_lo = 0, cull_y_hi = 0;
    // ... and whole workgroups of build_records (256 consecutive cells) are judged first by a sphere about their cells
    // (centre x, y, z and radius in the grid's own coordinates, made at upload; the view is a rigid motion): a context that
    // renders an eighth of the rows does not look at the vertices of the other seven eighths' cells.  nullptr: not judged.
    const double4* block_sphere = nullptr;
    RotationList rot{};  // the frame's view (for the spheres' centres)
    SplitParams split{};  // "depth_split": build_records lists the cells that straddle a cutting plane (n_slabs > 1)
    // boundary-face records for entry_raster_rec (nullptr: none wanted), stamped bf_seq; which faces a ray enters through
    BFaceRecord* bfrec = nullptr;
    uint32_t bf_seq = 0;
    int32_t bf_want_upper = 0;
    double bf_key_slack = 0.0;
};

struct WalkParams {
    const ExitRecord* xrec;
    EntryHead* entry_head;      // [n_local_px] entries of the pixel + overflow chain; the walk hands it back zeroed
    const Entry* entry_first;   // [n_local_px] first entry (valid where entry_count > 0)
    const Entry* entry_pool;    // overflow entries, chained from entry_first[].next
    int64_t pool_capacity;
    double key_slack;           // the frame's uniform entry-key slack (walk_common.hpp: next_entry); 0: keys = depths
    const uint32_t* mask;       // [n_local_px] or nullptr
    SolidTable solids;
    const double* Xtab;
    const double* Ytab;
    float2* out;                // [n_local_rows][res_x]
    ImageParams im;
    double t_cutoff;            // front-to-back early-out on transmittance
    uint32_t max_steps;
    int32_t xcd_mode;
    int32_t band_tiles;         // xcd_mode 1: workgroup-tile rows per band (set by launch_walk)
    int32_t lds_stage;          // 2: walk_composite_lds with LDS-DMA staging, 1: staged through registers, 0: direct loads
    int32_t stage_slots;        // lds_stage 1 / 2: 14 or 21 distinct cells staged per wavefront and step (2 / 3 DMA passes of 7)
    int32_t band_rows;          // xcd_mode 1: image rows per band (0: 32)
    int32_t lds_pad;            // tuning: extra dynamic LDS per workgroup (bytes) to cap the resident wavefronts
    int32_t order;              // 0: back to front in the reference's arithmetic; 1: front to back + early-out
    FrameCounters* counters;
    // tiles that hold an entry carry tile_stamp (walk_common.hpp: RasterArgs::tile_flag); nullptr: every tile reads its heads
    const uint32_t* tile_flag;
    uint32_t tile_stamp;
    // 1: no step of this grid can give exp an argument beyond -1/8, whatever the view: min(alpha limit, largest alpha)
    // times the longest edge of any cell stays below it (c_api.hip).  The walk then runs the instantiation that holds
    // only the short exp series — neither the wave-wide test of the argument per step nor the general exp's code
    int32_t small_exp_only;
    // rows of super-blocks (xcd_mode 2) start in the order of their cost in an earlier frame, dearest first (a launch
    // ends with its last wavefronts: let those be short ones): sb_order[k] = the k-th row to start, n_sb_rows of
    // them (0: image order); every wavefront adds its segments to sb_cost[its row] (nullptr: not collected).  The
    // order is worked out by the host whenever it waits for a frame anyway (c_api.hip), and travels as a kernel argument.
    uint32_t* sb_cost;
    int32_t n_sb_rows;
    uint8_t sb_order[128];
    uint32_t* row_cost;         // [n_local_rows] segments per row of this frame, or nullptr
    // 1: the walk leaves the per-pixel entry heads as they are (it normally hands them back cleared): the next frame has
    // the same view and reuses the entry lists, the records and the transformed vertices (c_api.hip: "view_cache")
    int32_t keep_entries;
    SplitParams split;          // "depth_split" (device_types.hpp); n_slabs <= 1: whole rays
    unsigned* sticky;           // [0] entries without a pool slot, [1] rays over the step bound, [2] rays that skipped an entry
                                // (interpenetrating components), summed over the frames since the host last looked; reset by the host only
};

// exact_kernels.hip (-ffp-contract=off)
// counters_to_clear: the frame's FrameCounters[kCounterShards], zeroed by the same launch (or nullptr)
// sb_cost_to_clear: the walk's per-row costs (WalkParams::sb_cost), n_sb of them, cleared by the same launch
// A frame that reuses the per-view data of the frame before: only what the walk adds to is cleared (the raster's part of
// the counters - pool_used, entry_overflow - stays), and the walk's per-row costs
// raster_from: where the raster's share of those counters stands, if not in `counters` itself (nullptr: it does)
void launch_clear_walk_counters(hipStream_t s, FrameCounters* counters, uint32_t* sb_cost_to_clear, int n_sb,
                                const FrameCounters* raster_from = nullptr);
void launch_transform_soa(hipStream_t s, const double* px, const double* py, const double* pz,
                          double* vx, double* vy, double* vz, int64_t n, const RotationList& R,
                          FrameCounters* counters_to_clear, uint32_t* sb_cost_to_clear = nullptr, int n_sb = 0);
constexpr int kMaxSbRows = 128;
// image rows per super-block row of the walk's xcd_mode 2 launch
int walk_sb_rows(int tile_shape, int band_rows);
void launch_transform_aos(hipStream_t s, const double* in, double* out, int64_t n,
                          const RotationList& R);
void launch_solid_mask_raster(hipStream_t s, const double* pts, const int4* faces, int64_t n_faces,
                              uint32_t value, const double* Ytab, const ImageParams& im, uint32_t* mask,
                              int lanes_per_face,  // power of two, 1..64: threads sharing the rows of one face
                              bool only_across_border = false);  // leave out faces a pixel inside the domain on every side

// bin-sort-resolve (exact_kernels.hip): the reference's algorithm on the GPU
void launch_bin_count(hipStream_t s, const GridView& g, const double* Xtab, const double* Ytab,
                      const ImageParams& im, int32_t* count, unsigned* odd_pixels);
void launch_bin_fill(hipStream_t s, const GridView& g, const double* Xtab, const double* Ytab,
                     const ImageParams& im, int32_t* count, const int64_t* offs, void* segs);
void launch_scan64(hipStream_t s, const int32_t* count, int64_t* offs, int64_t n, int64_t* scratch);
void launch_resolve(hipStream_t s, const GridView& g, const ImageParams& im, const int64_t* offs, void* segs,
                    const uint32_t* mask, const SolidTable& solids, double alpha_limit, float2* out,
                    FrameCounters* counters);
size_t segment_bytes();
// dst[p] = src[p] wherever src[p] != 0 (a cached solid mask laid over the frame's mask, in slot order)
void launch_mask_overlay(hipStream_t s, const uint32_t* src, uint32_t* dst, int64_t n_padded);

// walk_kernels.hip
// key_slack of the two launchers below: how far an entry's depth key lies behind its face (walk_common.hpp:
// entry_key_slack) = this fraction of the diagonal of the grid's bounding box (+ a term for the rounding of an
// absolute depth, c_api.hip), the same for every face of a frame
constexpr double kEntryKeySlack = 0x1p-24;
void launch_build_records(hipStream_t s, const GridView& g, double alpha_limit, int order);
// (g.bfrec set: from the face records build_records left - entry_raster_rec; else every face finds its own vertices)
void launch_entry_lists(hipStream_t s, const GridView& g, const double* Xtab, const double* Ytab,
                        const ImageParams& im, EntryHead* head, Entry* first, Entry* pool, int64_t capacity,
                        FrameCounters* counters, unsigned* sticky, int want_upper, double key_slack,
                        uint32_t* tile_flag = nullptr, uint32_t tile_stamp = 0);
// build_records and entry_raster as one launch of interleaved workgroups
void launch_setup_fused(hipStream_t s, const GridView& g, double alpha_limit, int order, const double* Xtab, const double* Ytab,
                        const ImageParams& im, EntryHead* head, Entry* first, Entry* pool, int64_t capacity,
                        FrameCounters* counters, unsigned* sticky, int want_upper, double key_slack);
void launch_walk(hipStream_t s, const WalkParams& p, int tile_shape);
// "depth_split": the cells build_records listed as straddling a cutting plane, scan-converted into split.plane_cell
void launch_plane_raster(hipStream_t s, const GridView& g, const double* Xtab, const double* Ytab, const ImageParams& im);
// pixels x, y tiles of the walk's 8x8 tiling (what SplitParams::arrivals / part_* are sized by)
int64_t walk_tiles(const ImageParams& im);

}  // namespace c5
