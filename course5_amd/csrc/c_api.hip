// C ABI (include/course5_hip.h) of the MI355X render path: context, device buffers and the
// per-frame kernel sequence.  No exceptions cross the ABI; every entry point returns a status
// and leaves a message for c5_last_error().
#include <hip/hip_runtime.h>

#include <algorithm>

#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/course5_hip.h"
#include "adjacency.hpp"
#include "device_types.hpp"
#include "kernels.hpp"

namespace {

std::string g_create_error;

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t want) {
        if (want <= bytes && ptr) return hipSuccess;
        if (ptr) {
            hipError_t e = hipFree(ptr);
            ptr = nullptr;
            bytes = 0;
            if (e != hipSuccess) return e;
        }
        if (want == 0) return hipSuccess;
        hipError_t e = hipMalloc(&ptr, want);
        if (e == hipSuccess) bytes = want;
        return e;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
    template <class T>
    T* as() const {
        return static_cast<T*>(ptr);
    }
};


struct Solid {
    int64_t n_tets = 0;      // as given
    int64_t n_points = 0;    // unique points
    int64_t n_faces = 0;     // unique faces
    int64_t n_interior = 0;  // of them, at the head of the list: faces with a cell on either side (adjacency.hpp)
    double colour = 0.0;
    // The faces in groups of about the same size (binary exponent of the longest edge, longest first; the interior faces'
    // groups before the others'): solid_mask_raster gives a face as many lanes as its rows ask for, and what a launch
    // lasts is the longest chain of rows ONE lane walks - the 130 000 surface triangles of the Roche lobe are a few pixels
    // each, its few hundred closing faces (object3d_base.cpp:171-174) span the whole solid.
    struct FaceGroup {
        int64_t begin, count;
        double longest;  // edge, object units (no rotation makes a face taller)
    };
    std::vector<FaceGroup> groups;
    double centre[3] = {0, 0, 0};        // bounding sphere of the points (object space)
    double radius = 0.0;
    DeviceBuffer raw;        // unique points [m][3]
    DeviceBuffer faces;      // unique faces, int4 (a, b, c, 0)
    DeviceBuffer view[2];    // transformed points, one per frame slot
    c5::RotationList rots{};
    // A solid whose view and image did not change since the frame before (the accretor sphere never rotates,
    // main.cpp:116; in a -D sweep only the lobe moves) is rastered ONCE into a mask of its own, which later
    // frames lay over theirs (one pass over the image instead of a transform and a raster of 10^5-10^6 faces).
    DeviceBuffer own_mask;
    c5::RotationList seen_rots{};
    c5::ImageParams seen_im{};
    uint64_t generation = 0, seen_generation = ~uint64_t{0};
    hipStream_t seen_stream = nullptr;  // the own mask is written and read in the order of ONE stream
    int unchanged_frames = 0;
    bool own_mask_ready = false;
};

bool same_rotations(const c5::RotationList& a, const c5::RotationList& b) {
    if (a.n != b.n) return false;
    for (int k = 0; k < a.n; ++k)
        if (a.axis[k] != b.axis[k] || a.cosv[k] != b.cosv[k] || a.sinv[k] != b.sinv[k] || a.x0[k] != b.x0[k]) return false;
    return true;
}
bool same_image(const c5::ImageParams& a, const c5::ImageParams& b) {
    return a.res_x == b.res_x && a.res_y == b.res_y && a.n_local_rows == b.n_local_rows && a.tile_rows == b.tile_rows &&
           a.rank == b.rank && a.world == b.world && a.row_begin == b.row_begin && a.row_count == b.row_count &&
           a.x_min == b.x_min && a.y_min == b.y_min && a.step_x == b.step_x && a.step_y == b.step_y;
}

constexpr int kWalkEventPool = 512;
constexpr size_t kStageChunk = size_t{4} << 20;  // pinned staging for pageable destinations, two of these
constexpr int kFrameSlots = 2;
constexpr int kStickyWords = 3;   // entries without a pool slot, rays over the step bound, rays that skipped an entry
constexpr int kStatusWords = 3;   // a frame's own walk_overflow, entry_overflow, overlap_rays (FrameCounters, shard 0)
constexpr size_t kCountersBytes = sizeof(c5::FrameCounters) * c5::kCounterLines;  // the shards + DepthFitSums (device_types.hpp)

// Everything one frame writes before its image: two slots, so that the per-view setup of frame
// k + 1 (HBM-bound: transform, records, entry lists, solid mask) can run on the auxiliary stream
// while walk_composite of frame k (VALU / address-path bound) runs on the main stream.
struct FrameSlot {
    DeviceBuffer vx, vy, vz, rec, count, head, first, pool, mask, counters, row_cost;
    DeviceBuffer sb;        // cost of the walk's rows of super-blocks in the last frame (WalkParams::sb_cost)
    long long sb_key = -1;  // the tiling they belong to (-1: not collected)
    int sb_n = 0;
    // "view_cache": the per-view data in this slot (transformed vertices, records, entry lists) were built for ...
    uint64_t setup_epoch = 0;     // ... this state of the context (c5_context::setup_epoch; 0: nothing built)
    c5::RotationList setup_view{};
    double setup_limit = 0.0;
    int setup_order = 0;
    bool setup_kept = false;      // ... and the walk that used them left the entry heads in place
    bool setup_reused = false;    // the last frame enqueued into this slot skipped the per-view setup
    const c5::FrameCounters* raster_counters = nullptr;  // device: counters of the frame whose raster built the slot's entry lists
    // "depth_split" (device_types.hpp: SplitParams)
    DeviceBuffer plane_cell, straddle, straddle_count, partials, arrivals;
    DeviceBuffer bfrec;          // BFaceRecord per boundary face (build_records -> entry_raster_rec), stamped with ...
    uint32_t bf_seq = 0;         // ... the number of the frame that wrote it
    int split_k = 0;             // slabs the buffers above are laid out for (0: none)
    int64_t split_px = 0, split_tiles = 0, split_cells = 0;
    uint64_t split_seq = 0;      // raster frames so far: stamp = seq % 15 + 1, counter half = seq & 1
    int setup_split = 1;         // slabs the slot's per-view data (plane cells) were built for ("view_cache")
    double setup_w[c5::kMaxSlabs + 1] = {};
    double setup_g[2] = {0.0, 0.0};
    int64_t entry_capacity = 0;
    // which 8x8 tiles hold an entry (walk_common.hpp: RasterArgs::tile_flag): a word per tile = the number of the raster run
    // that found one there
    DeviceBuffer tile_flag;
    uint32_t flag_seq = 0;
    int64_t flag_tiles = 0;
    bool flags_valid = false;  // the slot's entry lists were built by a raster that kept them
    bool head_clean = false;  // the per-pixel entry heads are all zero (the walk kernels leave them so)
    c5::FrameCounters* host_counters = nullptr;  // pinned
    hipEvent_t setup_done = nullptr, walk_done = nullptr;
    bool walk_recorded = false;
    hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

}  // namespace

struct c5_context {
    int device = 0;
    hipStream_t stream = nullptr;      // stream in use
    hipStream_t own_stream = nullptr;  // created by c5_create
    std::string error;

    // persistent grid
    int64_t n_pts = 0, n_cells = 0, n_bfaces = 0;
    hipStream_t aux_stream = nullptr;  // per-view setup of the next frame
    hipStream_t side_stream = nullptr; // entry lists + solid mask beside build_records
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    int fuse_setup = 0;     // build_records + entry_raster as one launch of interleaved workgroups: measured 0.119 ms against
                            // 0.047 + 0.047 ms for the two launches on the C3 frame (the raster inherits the records' 49 KB of LDS)
    int cell_order = 1;     // "cell_order": c5_upload_grid keeps the cells in Morton order of their centroids (set BEFORE the upload)
    std::vector<int32_t> cell_perm;  // device index -> the caller's (empty: the same)
    int block_cull = 1;     // "block_cull": build_records judges whole workgroups by a sphere about their cells (a part of the rows only)
    int tile_flags = 1;     // "tile_flags": the raster marks the tiles that hold an entry, the walk looks there first (enqueue_frame)
    int cost_order = 1;     // "cost_order": rows of super-blocks start dearest first (by the last frame the host waited for)
    uint8_t sb_order[128] = {};
    long long sb_order_key = -1;
    int sb_order_n = 0;
    uint32_t* host_sb = nullptr;  // pinned: the last frame's per-row costs
    int entry_key = 1;      // "entry_key": 1 = entries keyed a slack behind their face (hanging-node interfaces), 0 = at the face (testing)
    int stage_slots = 0;    // "stage_slots": 0 = chosen per frame from rays_per_cell, or 14 / 21
    double rays_per_cell = 0.0;  // of the last finished frame (0: none yet)
    int solid_cache = 1;    // a solid unchanged since the frame before is not rastered again (enqueue_solids)
    int view_cache = 1;     // "view_cache": a frame with the view of the two before it reuses their per-view data (enqueue_frame)
    uint64_t setup_epoch = 1;  // bumped by everything but the view, the alpha limit and the solids that the per-view data depend on
    int solid_interior_faces = 0;  // 1: interior faces are rastered too (they cover nothing the others do not; testing)
    int depth_split = 0;    // "depth_split": 0 = chosen per frame (split_auto_k), 1 = never, 2..8 = that many slabs
    int entry_records = 1;  // "entry_records": build_records leaves a record per boundary face for the entry raster (0: the raster gathers)
    int split_auto_k = 1;   // what the last finished frame suggests (finish_frame)
    bool ray_depth_known = false;  // ... and the depths its rays ran between (walk coordinate)
    double ray_depth_lo = 0.0, ray_depth_hi = 0.0;
    // ... and the planes fitted through where its (sampled) rays entered the grid and where they ended (DepthFitSums):
    // common tilt (fit_gx, fit_gy) and the two planes' depths at x = y = 0 once that tilt is taken out
    bool fit_known = false;
    double fit_gx = 0.0, fit_gy = 0.0, fit_entry0 = 0.0, fit_exit0 = 0.0;
    double split_tilt_x = 0.0, split_tilt_y = 0.0;  // "split_tilt_x" / "_y" (testing): the tilt of a FORCED split's planes
    double box_lo[3] = {0, 0, 0}, box_hi[3] = {0, 0, 0};  // the grid's bounding box in object space
    double alpha_floor = 0.0;  // smallest alpha of the grid that is >= DBL_EPSILON (+inf: none)
    int overlap_setup = 0;  // measured: 1.27 vs 1.26 ms/frame, the side stream buys nothing
    DeviceBuffer px, py, pz, cell_vert, cell_adj, alpha, q, bface, block_sphere;
    double alpha_top = 0.0;      // largest alpha of the grid (c5_upload_grid / c5_update_scalars)
    double edge_max = 0.0;       // longest edge of any cell: no view makes a cell longer along a ray
    double grid_diagonal = 0.0;  // of the grid's bounding box in object space: no rotation makes the grid longer along a ray
    double coord_max = 0.0;      // largest |coordinate| (what the rounding of an absolute depth scales with)
    FrameSlot slots[kFrameSlots];
    int64_t frame_index = 0;
    int last_slot = 0;
    void* last_counters = nullptr;  // device: the counters the last enqueued frame adds to (the slot's, or a host-ring frame's own)
    int row_cost_slot = 0;          // the slot whose row_cost[] holds the segments per row of the last frame that counted them
    int algorithm = 0;        // 0: walk, 1: bin_sort_resolve
    bool grid_conforming = true;   // no face in more than two cells (c5_upload_grid)
    bool overlap_seen = false;     // a frame's walk met interpenetrating components (finish_frame): bin_sort_resolve until the next upload
    DeviceBuffer offs64, scratch64, segs;  // bin_sort_resolve
    int pipeline = 0;  // measured at the end of round 1: 0.689 ms per C3 frame with it, 0.702 without
    c5::RotationList view{};
    Solid solids[C5_MAX_SOLIDS];

    // image
    bool have_image = false;
    double bounds[4] = {0, 0, 0, 0};
    c5::ImageParams im{};
    int cfg_tile_rows = 0, cfg_rank = 0, cfg_world = 1;
    int cfg_row_begin = 0, cfg_row_count = -1;  // -1: all rows
    DeviceBuffer xtab, ytab, out, sticky;  // sticky: kStickyWords x u32 failure words that persist across frames (kernels.hpp: WalkParams::sticky)
    unsigned* host_sticky = nullptr;       // pinned copy, refreshed at the end of every frame
    std::vector<double> host_ytab;
    int row_costs = 0;
    bool row_costs_collected = false;  // some frame since the rows were last laid out counted its segments per row

    // options
    double alpha_limit = 2.5;
    double t_cutoff = 1e-12;
    int tile_shape = 3;  // 8x8 pixels per wavefront (fewest distinct cells per step), one wavefront per workgroup
    int xcd_mode = 2;
    int lds_pad = 0;
    int band_rows = 0;
    int order = 0;
    int lds_stage = 2;
    // instruments, off unless asked for: the six stage events cost 21-25 us of a 0.52-ms frame when frames follow one another
    // without a wait, the two around the walk 6 (profiles/experiments.md)
    int stage_timing = 0;
    int walk_timing = 0;
    unsigned walk_seq = 0;

    // events
    hipEvent_t walk_a[kWalkEventPool];
    hipEvent_t walk_b[kWalkEventPool];
    int walk_used = 0;
    double walk_ms_sum = 0.0;
    int64_t walk_launches = 0;

    // frames delivered to host memory (c5_render_host_async / _wait, c5_render)
    struct HostFrame {
        DeviceBuffer img;
        // statistics of THIS frame (FrameCounters[kCounterShards]): frames in flight never share them, so the frame's own
        // failure words can be read behind it whatever the frames after it are doing
        DeviceBuffer counters;
        hipEvent_t rendered = nullptr, copied = nullptr;
        unsigned* status = nullptr;  // pinned: {rays over the step bound, entries without a pool slot} of THIS frame
    };
    hipStream_t copy_stream = nullptr;
    HostFrame hring[C5_HOST_RING];
    int hr_head = 0, hr_count = 0, hr_next = 0;
    int hr_retry_left = 0;  // outstanding frames that were enqueued before an overflow was noticed
    void* stage[2] = {nullptr, nullptr};  // pinned staging chunks for pageable destinations
    hipEvent_t stage_ev[2] = {nullptr, nullptr};

    bool using_caller_stream = false;
    bool frame_pending = false;
    bool counters_on_host = true;  // the last frame's counters / sticky words have been copied to the host
    bool frame_timed = false;
    c5_stats last{};
};

namespace {

int fail(c5_context* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->error = buf;
    else
        g_create_error = buf;
    return code;
}

#define C5_HIP(ctx, expr)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail((ctx), C5_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                     \
    } while (0)

int bind_device(c5_context* ctx) {
    C5_HIP(ctx, hipSetDevice(ctx->device));
    return C5_OK;
}

int to_rotation_list(c5_context* ctx, const c5_rotation* rots, int n, c5::RotationList& out) {
    if (n < 0 || n > C5_MAX_ROTATIONS) return fail(ctx, C5_ERR_INVALID, "rotation count %d out of range", n);
    if (n > 0 && !rots) return fail(ctx, C5_ERR_INVALID, "null rotation list");
    out.n = n;
    for (int k = 0; k < n; ++k) {
        if (rots[k].axis != 0 && rots[k].axis != 1)
            return fail(ctx, C5_ERR_INVALID, "rotation axis must be 0 (x) or 1 (y)");
        out.axis[k] = rots[k].axis;
        // libm on the host, like tetra.cpp:46-47,58-59
        out.cosv[k] = std::cos(rots[k].angle);
        out.sinv[k] = std::sin(rots[k].angle);
        out.x0[k] = rots[k].x0;
    }
    return C5_OK;
}

// tetra.cpp:44-62 on the host (what rotate_point does on the device), for bounding boxes and centres
void rotate_host(const c5::RotationList& R, double c[3]) {
    for (int r = 0; r < R.n; ++r) {
        const double co = R.cosv[r], si = R.sinv[r];
        if (R.axis[r] == 0) {
            const double y_old = c[1];
            c[1] = c[1] * co - c[2] * si;
            c[2] = y_old * si + c[2] * co;
        } else {
            c[0] -= R.x0[r];
            const double x_old = c[0];
            c[0] = c[0] * co - c[2] * si;
            c[2] = x_old * si + c[2] * co;
            c[0] += R.x0[r];
        }
    }
}

// "depth_split": below this clamped alpha (and above DBL_EPSILON, where a cell stops taking part: line.cpp:220-224) the
// reference's recurrence I = (Q - (Q - alpha I) exp(-alpha dz)) / alpha is dominated by its own cancellation error
// (eps Q / alpha per step: golden fixture g4), which depends on the very bits of the I it is fed: partial integrals
// composed afterwards cannot reproduce it.  A grid with such a cell is walked whole unless the caller forces the split.
constexpr double kSplitAlphaFloor = 1e-6;

int recompute_rows(c5_context* ctx) {
    c5::ImageParams& im = ctx->im;
    im.tile_rows = (ctx->cfg_tile_rows > 0) ? ctx->cfg_tile_rows : (im.res_y > 0 ? im.res_y : 1);
    im.rank = ctx->cfg_rank;
    im.world = ctx->cfg_world;
    im.row_begin = ctx->cfg_row_begin;
    im.row_count = (ctx->cfg_row_count < 0) ? im.res_y - im.row_begin : ctx->cfg_row_count;
    if (im.row_begin < 0 || im.row_count < 0 || im.row_begin + im.row_count > im.res_y)
        return fail(ctx, C5_ERR_INVALID, "row range [%d, %d) outside the image of %d rows", im.row_begin,
                    im.row_begin + im.row_count, im.res_y);
    int n = 0;
    for (int r = 0; r < im.res_y; ++r)
        if (c5::local_row_of(im, r) >= 0) ++n;
    im.n_local_rows = n;
    ctx->row_costs_collected = false;
    return C5_OK;
}

int ensure_image_buffers(c5_context* ctx) {
    const c5::ImageParams& im = ctx->im;
    const int64_t n_px = static_cast<int64_t>(im.n_local_rows) * im.res_x;
    const int64_t padded = ((n_px + 1023) / 1024) * 1024;
    C5_HIP(ctx, ctx->out.ensure(static_cast<size_t>(padded) * sizeof(float) * 2));
    for (int k = 0; k < (ctx->pipeline ? kFrameSlots : 1); ++k) {
        FrameSlot& fs = ctx->slots[k];
        C5_HIP(ctx, fs.count.ensure(static_cast<size_t>(padded + 1024) * sizeof(int32_t)));
        C5_HIP(ctx, fs.head.ensure(static_cast<size_t>(padded) * sizeof(c5::EntryHead)));
        fs.head_clean = false;
        C5_HIP(ctx, fs.first.ensure(static_cast<size_t>(padded) * sizeof(c5::Entry)));
        C5_HIP(ctx, fs.mask.ensure(static_cast<size_t>(padded) * sizeof(uint32_t)));
        C5_HIP(ctx, fs.row_cost.ensure(static_cast<size_t>(im.n_local_rows + 64) * sizeof(uint32_t)));
        // overflow pool: second and further entries of a ray only.  A frame overflows iff its total demand
        // exceeds the capacity (entry_raster); finish_frame keeps the capacity at twice the demand of the
        // last frame it has seen, so only a jump of the demand between two looks can cost a C5_RETRY.
        if (fs.entry_capacity < n_px / 4 + 8192) {
            fs.entry_capacity = n_px / 4 + 8192;
            C5_HIP(ctx, fs.pool.ensure(static_cast<size_t>(fs.entry_capacity) * sizeof(c5::Entry)));
            ++ctx->setup_epoch;
        }
    }
    return C5_OK;
}

// Wait until nothing of this context is running (both streams).
int drain(c5_context* ctx) {
    C5_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->aux_stream) C5_HIP(ctx, hipStreamSynchronize(ctx->aux_stream));
    if (ctx->side_stream) C5_HIP(ctx, hipStreamSynchronize(ctx->side_stream));
    return C5_OK;
}

// (a9) solid mask of this frame into fs.mask; returns whether any solid exists
int enqueue_solids(c5_context* ctx, FrameSlot& fs, int slot_id, hipStream_t s, c5::SolidTable& table, bool& any_solid) {
    const c5::ImageParams& im = ctx->im;
    const int64_t n_px = static_cast<int64_t>(im.n_local_rows) * im.res_x;
    const int64_t padded = ((n_px + 1023) / 1024) * 1024;
    any_solid = false;
    for (int k = 0; k < C5_MAX_SOLIDS; ++k) {
        table.colour[k] = ctx->solids[k].colour;
        if (ctx->solids[k].n_tets > 0) any_solid = true;
    }
    table.n_slots = C5_MAX_SOLIDS;
    if (any_solid) {
        C5_HIP(ctx, hipMemsetAsync(fs.mask.ptr, 0, static_cast<size_t>(padded) * sizeof(uint32_t), s));
        for (int k = 0; k < C5_MAX_SOLIDS; ++k) {
            Solid& so = ctx->solids[k];
            if (so.n_tets <= 0) continue;
            // unchanged since the frame before, and still on the same stream?
            const bool same = ctx->solid_cache && !ctx->pipeline && so.seen_stream == s && so.seen_generation == so.generation &&
                              same_rotations(so.seen_rots, so.rots) && same_image(so.seen_im, im);
            so.unchanged_frames = same ? so.unchanged_frames + 1 : 0;
            if (!same) {
                so.own_mask_ready = false;
                so.seen_rots = so.rots;
                so.seen_im = im;
                so.seen_generation = so.generation;
                so.seen_stream = s;
            }
            if (so.own_mask_ready) {
                c5::launch_mask_overlay(s, so.own_mask.as<uint32_t>(), fs.mask.as<uint32_t>(), padded);
                continue;
            }
            uint32_t* target = fs.mask.as<uint32_t>();
            if (so.unchanged_frames >= 1) {  // second frame in a row with this view: raster into the solid's own mask
                C5_HIP(ctx, so.own_mask.ensure(static_cast<size_t>(padded) * sizeof(uint32_t)));
                C5_HIP(ctx, hipMemsetAsync(so.own_mask.ptr, 0, static_cast<size_t>(padded) * sizeof(uint32_t), s));
                target = so.own_mask.as<uint32_t>();
            }
            c5::launch_transform_aos(s, so.raw.as<double>(), so.view[slot_id].as<double>(), so.n_points, so.rots);
            // Interior faces (a cell on either side: they cover no pixel the solid's other faces do not) are skipped -
            // where get_pixel_by_x/_y's clamp has no hand in their pixels (plane.cpp:194-212: what a face beyond a
            // border smears onto it is NOT the projection of anything).  A solid whose bounding sphere lies inside
            // the domain: none of them is even launched; one that reaches a border: each interior face decides for
            // itself (solid_mask_raster, only_across_border).
            int64_t skip = 0;
            const bool filter_interior = !ctx->solid_interior_faces && so.n_interior > 0 && so.n_interior < so.n_faces;
            if (filter_interior) {
                double c[3] = {so.centre[0], so.centre[1], so.centre[2]};
                for (int r = 0; r < so.rots.n; ++r) {  // tetra.cpp:44-62, as rotate_point on the device
                    const double co = so.rots.cosv[r], si = so.rots.sinv[r];
                    if (so.rots.axis[r] == 0) {
                        const double y_old = c[1];
                        c[1] = c[1] * co - c[2] * si;
                        c[2] = y_old * si + c[2] * co;
                    } else {
                        c[0] -= so.rots.x0[r];
                        const double x_old = c[0];
                        c[0] = c[0] * co - c[2] * si;
                        c[2] = x_old * si + c[2] * co;
                        c[0] += so.rots.x0[r];
                    }
                }
                const double pad = so.radius * (1.0 + 1e-9) + 2.0 * std::fmax(std::fabs(im.step_x), std::fabs(im.step_y));
                const double x_a = im.x_min, x_b = im.x_min + im.step_x * (im.res_x - 1);
                const double y_a = im.y_min, y_b = im.y_min + im.step_y * (im.res_y - 1);
                if (c[0] - pad > std::fmin(x_a, x_b) && c[0] + pad < std::fmax(x_a, x_b) && c[1] - pad > std::fmin(y_a, y_b) &&
                    c[1] + pad < std::fmax(y_a, y_b))
                    skip = so.n_interior;
            }
            // lanes per face: about one per eight image rows of the group's tallest face, a power of two up to 64;
            // neighbouring groups that come to the same number share a launch
            int64_t run_begin = -1, run_count = 0;
            int run_lanes = 0;
            bool run_filtered = false;
            auto flush = [&]() {
                if (run_count > 0)
                    c5::launch_solid_mask_raster(s, so.view[slot_id].as<double>(), so.faces.as<int4>() + run_begin, run_count,
                                                 static_cast<uint32_t>(k) + 1u, ctx->ytab.as<double>(), im, target, run_lanes,
                                                 run_filtered);
                run_count = 0;
            };
            for (const Solid::FaceGroup& g : so.groups) {
                if (g.begin < skip) continue;  // an interior group of a solid inside the domain
                const bool filtered = filter_interior && g.begin < so.n_interior;
                // a lane per eight rows of the group's tallest face while the group alone does not fill the GPU, per
                // thirty-two once it does (every lane of a face repeats the face's set-up)
                const double rows = g.longest / std::fabs(im.step_y);
                int lanes = 1;
                while (lanes < 64 && rows > 8.0 * lanes && (g.count * lanes < (int64_t{1} << 18) || rows > 32.0 * lanes)) lanes *= 2;
                if (run_count > 0 && (lanes != run_lanes || filtered != run_filtered || g.begin != run_begin + run_count)) flush();
                if (run_count == 0) run_begin = g.begin, run_lanes = lanes, run_filtered = filtered;
                run_count += g.count;
            }
            flush();
            if (target != fs.mask.as<uint32_t>()) {
                so.own_mask_ready = true;
                c5::launch_mask_overlay(s, so.own_mask.as<uint32_t>(), fs.mask.as<uint32_t>(), padded);
            }
        }
    }
    return C5_OK;
}

// bin_sort_resolve: the reference's algorithm (plane.cpp:184-192 + 144-172) on the GPU.  Needs the
// segment total on the host between its two binning passes, so it synchronises.
int enqueue_bin_sort(c5_context* ctx, FrameSlot& fs, const c5::GridView& g, int slot_id, float2* out_dev,
                     hipStream_t s, hipStream_t main_s, bool timed, c5::FrameCounters* counters) {
    const c5::ImageParams& im = ctx->im;
    const int64_t n_px = static_cast<int64_t>(im.n_local_rows) * im.res_x;
    const int64_t padded = ((n_px + 1023) / 1024) * 1024;
    auto mark = [&](int k, hipStream_t st) -> hipError_t { return timed ? hipEventRecord(fs.ev[k], st) : hipSuccess; };
    C5_HIP(ctx, mark(2, s));
    C5_HIP(ctx, ctx->offs64.ensure(static_cast<size_t>(padded + 1024) * sizeof(int64_t)));
    C5_HIP(ctx, ctx->scratch64.ensure(static_cast<size_t>(padded / 1024 + 1024) * sizeof(int64_t)));
    C5_HIP(ctx, hipMemsetAsync(fs.count.ptr, 0, static_cast<size_t>(padded + 1) * sizeof(int32_t), s));
    unsigned* odd = &counters->odd_pixels;
    c5::launch_bin_count(s, g, ctx->xtab.as<double>(), ctx->ytab.as<double>(), im, fs.count.as<int32_t>(), odd);
    c5::launch_scan64(s, fs.count.as<int32_t>(), ctx->offs64.as<int64_t>(), n_px, ctx->scratch64.as<int64_t>());
    int64_t total = 0;
    C5_HIP(ctx, hipMemcpyAsync(&total, ctx->offs64.as<int64_t>() + n_px, sizeof total, hipMemcpyDeviceToHost, s));
    C5_HIP(ctx, hipStreamSynchronize(s));
    C5_HIP(ctx, ctx->segs.ensure(static_cast<size_t>(total + 16) * c5::segment_bytes()));
    c5::launch_bin_fill(s, g, ctx->xtab.as<double>(), ctx->ytab.as<double>(), im, fs.count.as<int32_t>(),
                        ctx->offs64.as<int64_t>(), ctx->segs.ptr);
    C5_HIP(ctx, mark(3, s));
    c5::SolidTable table{};
    bool any_solid = false;
    int rc = enqueue_solids(ctx, fs, slot_id, s, table, any_solid);
    if (rc) return rc;
    C5_HIP(ctx, mark(4, s));
    c5::launch_resolve(s, g, im, ctx->offs64.as<int64_t>(), ctx->segs.ptr, any_solid ? fs.mask.as<uint32_t>() : nullptr,
                       table, ctx->alpha_limit, out_dev, counters);
    C5_HIP(ctx, mark(5, s));
    C5_HIP(ctx, hipGetLastError());
    C5_HIP(ctx, hipMemcpyAsync(fs.host_counters, counters, kCountersBytes, hipMemcpyDeviceToHost, s));
    C5_HIP(ctx, hipMemcpyAsync(ctx->host_sticky, ctx->sticky.ptr, kStickyWords * sizeof(unsigned), hipMemcpyDeviceToHost, s));
    C5_HIP(ctx, hipStreamSynchronize(s));
    ctx->counters_on_host = true;
    fs.host_counters->seg_tiles = static_cast<unsigned long long>(total);  // (fewer than 2^40: they were all allocated)
    if (ctx->pipeline) {  // keep the two-stream bookkeeping consistent
        C5_HIP(ctx, hipEventRecord(fs.setup_done, s));
        C5_HIP(ctx, hipStreamWaitEvent(main_s, fs.setup_done, 0));
        C5_HIP(ctx, hipEventRecord(fs.walk_done, main_s));
        fs.walk_recorded = true;
    }
    ctx->last_slot = slot_id;
    ctx->last_counters = counters;
    ctx->frame_index += 1;
    ctx->frame_pending = true;
    ctx->frame_timed = timed;
    return C5_OK;
}

// Enqueue one frame; the image goes to out_dev.  The per-view setup runs on the auxiliary stream
// into frame slot (frame_index & 1), the walk on the main stream once that setup is done, so the
// setup of the next frame overlaps this frame's walk.
// own_counters: the frame's statistics go to these FrameCounters[kCounterShards] instead of the slot's (frames delivered to
// host memory: each frame of the ring keeps its own, c5_render_host_async).
int enqueue_frame(c5_context* ctx, float2* out_dev, c5::FrameCounters* own_counters = nullptr) {
    if (ctx->n_cells <= 0 && [&] {
            for (const Solid& s : ctx->solids)
                if (s.n_tets > 0) return false;
            return true;
        }())
        return fail(ctx, C5_ERR_STATE, "plane initializer. empty set of objects to render");  // plane.cpp:269-271
    if (!ctx->have_image) return fail(ctx, C5_ERR_STATE, "critical error. empty plane");  // plane.cpp:151-153
    int rc = bind_device(ctx);
    if (rc) return rc;

    const int slot_id = ctx->pipeline ? static_cast<int>(ctx->frame_index & 1) : 0;
    FrameSlot& fs = ctx->slots[slot_id];
    hipStream_t main_s = ctx->stream;
    hipStream_t s = ctx->pipeline ? ctx->aux_stream : ctx->stream;  // setup stream
    const c5::ImageParams& im = ctx->im;
    const int64_t n_px = static_cast<int64_t>(im.n_local_rows) * im.res_x;
    const int64_t padded = ((n_px + 1023) / 1024) * 1024;
    const bool timed = ctx->stage_timing != 0;
    auto mark = [&](int k, hipStream_t st) -> hipError_t { return timed ? hipEventRecord(fs.ev[k], st) : hipSuccess; };
    c5::FrameCounters* const counters = own_counters ? own_counters : fs.counters.as<c5::FrameCounters>();

    // the slot's buffers are free once the walk that last read them has finished
    if (ctx->pipeline && fs.walk_recorded) C5_HIP(ctx, hipStreamWaitEvent(s, fs.walk_done, 0));

    C5_HIP(ctx, mark(0, s));

    c5::GridView g;
    g.n_pts = ctx->n_pts;
    g.n_cells = ctx->n_cells;
    g.n_bfaces = ctx->n_bfaces;
    g.px = ctx->px.as<double>();
    g.py = ctx->py.as<double>();
    g.pz = ctx->pz.as<double>();
    g.vx = fs.vx.as<double>();
    g.vy = fs.vy.as<double>();
    g.vz = fs.vz.as<double>();
    g.cell_vert = ctx->cell_vert.as<int4>();
    g.cell_adj = ctx->cell_adj.as<int4>();
    g.alpha = ctx->alpha.as<double>();
    g.q = ctx->q.as<double>();
    g.bface = ctx->bface.as<uint32_t>();
    g.xrec = fs.rec.as<c5::ExitRecord>();
    g.rot = ctx->view;
    // (whole workgroups of build_records judged by their cells' sphere: only where the rows are a part of the image)
    g.block_sphere = (ctx->block_cull && im.n_local_rows > 0 && im.n_local_rows < im.res_y && ctx->block_sphere.ptr) ? ctx->block_sphere.as<double4>() : nullptr;
    // y band of the rows this context renders (one pixel of slack on both sides)
    if (im.n_local_rows > 0) {
        const int first = c5::global_row_of(im, 0), last = c5::global_row_of(im, im.n_local_rows - 1);
        const double pad = std::fabs(im.step_y);
        const double ya = ctx->host_ytab[static_cast<size_t>(first)], yb = ctx->host_ytab[static_cast<size_t>(last)];
        g.cull_y_lo = std::fmin(ya, yb) - pad;
        g.cull_y_hi = std::fmax(ya, yb) + pad;
    } else {
        g.cull_y_lo = 1.0;
        g.cull_y_hi = -1.0;
    }

    // (a2) view transform
    // (a2) + the frame's statistics cleared by the same launch
    // + the walk's per-row costs cleared (what decides the order its rows of super-blocks start in, below)
    uint32_t* sb = nullptr;
    int n_sb = 0;
    long long sb_key = -1;
    if (ctx->cost_order && ctx->xcd_mode == 2 && ctx->lds_stage && im.n_local_rows > 0) {
        const int sb_rows = c5::walk_sb_rows(ctx->tile_shape, ctx->band_rows);
        n_sb = (im.n_local_rows + sb_rows - 1) / sb_rows;
        if (n_sb <= c5::kMaxSbRows) {
            if (!fs.sb.ptr) {
                C5_HIP(ctx, fs.sb.ensure(c5::kMaxSbRows * sizeof(uint32_t)));
                C5_HIP(ctx, hipMemsetAsync(fs.sb.ptr, 0, fs.sb.bytes, s));
            }
            sb_key = (static_cast<long long>(im.n_local_rows) << 20) ^ (static_cast<long long>(sb_rows) << 4) ^ ctx->tile_shape;
            sb = fs.sb.as<uint32_t>();
        } else {
            n_sb = 0;
        }
    }
    fs.sb_key = sb_key;
    fs.sb_n = n_sb;
    const bool bin_sort = ctx->algorithm == 1 || !ctx->grid_conforming || ctx->overlap_seen;
    // "depth_split" (device_types.hpp: SplitParams): how many slabs of depth this frame's rays are cut into
    int split_k = 1;
    double split_w[c5::kMaxSlabs + 1] = {};
    double split_g[2] = {0.0, 0.0};  // the planes' common tilt
    {
        const bool able = !bin_sort && !ctx->pipeline && !ctx->fuse_setup && !ctx->overlap_setup && ctx->order == 0 &&
                          ctx->tile_shape == 3 && ctx->xcd_mode == 2 && ctx->lds_stage == 2 && g.n_cells > 0 &&
                          g.n_cells < (int64_t{1} << 25) && im.n_local_rows > 0;
        double a_floor = ctx->alpha_floor;  // the smallest clamped alpha >= DBL_EPSILON any cell can have
        if (ctx->alpha_limit < a_floor) a_floor = ctx->alpha_limit >= DBL_EPSILON ? ctx->alpha_limit : INFINITY;
        const bool well_conditioned = a_floor >= kSplitAlphaFloor;
        int want = 1;
        if (ctx->depth_split >= 2) want = ctx->depth_split;  // (forced: the caller answers for the conditioning)
        else if (ctx->depth_split == 0 && well_conditioned) want = ctx->split_auto_k;
        if (able && want > 1) {
            // the grid's depth range under this view, from the corners of its bounding box; planes at equal distances
            double z_lo = INFINITY, z_hi = -INFINITY;
            for (int corner = 0; corner < 8; ++corner) {
                double c[3] = {(corner & 1) ? ctx->box_hi[0] : ctx->box_lo[0], (corner & 2) ? ctx->box_hi[1] : ctx->box_lo[1],
                               (corner & 4) ? ctx->box_hi[2] : ctx->box_lo[2]};
                rotate_host(ctx->view, c);
                z_lo = std::fmin(z_lo, c[2]);
                z_hi = std::fmax(z_hi, c[2]);
            }
            // (the rays of the last finished frame ran between narrower bounds: rows of a frame, a view from a corner)
            // (only when the library chooses the slabs itself: a forced count keeps planes that depend on the view alone, so
            // that renders of different rows of one frame stay bit-equal)
            if (ctx->depth_split == 0 && ctx->ray_depth_known && ctx->ray_depth_lo >= z_lo && ctx->ray_depth_hi <= z_hi)
                z_lo = ctx->ray_depth_lo, z_hi = ctx->ray_depth_hi;
            // Planes of constant depth cut the rays of an oblique view at different fractions (the cube of the benchmark seen
            // at -X 0.1 -Y 0.07: a ray's entry depth changes by 0.22 across the image, a fifth of its length - the longest
            // third of a ray cut in three had 83 of its 183 steps).  With the planes fitted through the last frame's entries
            // and ends at hand, the cutting planes take their mean tilt and divide the stretch between the two at the
            // sample's centre: parallel planes, in order everywhere.  Quantised (2^-16, 2^-20 of the grid's size), so that
            // the last bits of sums added in another order do not move them from frame to frame.
            if (ctx->depth_split == 0 && ctx->fit_known) {
                const double q_g = 0x1p-16, q_w = std::ldexp(std::fmax(ctx->grid_diagonal, 1e-300), -20);
                split_g[0] = std::round(ctx->fit_gx / q_g) * q_g;
                split_g[1] = std::round(ctx->fit_gy / q_g) * q_g;
                z_lo = std::round(ctx->fit_entry0 / q_w) * q_w;
                z_hi = std::round(ctx->fit_exit0 / q_w) * q_w;
            } else if (ctx->depth_split >= 2 && (ctx->split_tilt_x != 0.0 || ctx->split_tilt_y != 0.0)) {
                // (testing: a forced tilt; the bounding box's corners in the tilted coordinate)
                split_g[0] = ctx->split_tilt_x, split_g[1] = ctx->split_tilt_y;
                z_lo = INFINITY, z_hi = -INFINITY;
                for (int corner = 0; corner < 8; ++corner) {
                    double c[3] = {(corner & 1) ? ctx->box_hi[0] : ctx->box_lo[0], (corner & 2) ? ctx->box_hi[1] : ctx->box_lo[1],
                                   (corner & 4) ? ctx->box_hi[2] : ctx->box_lo[2]};
                    rotate_host(ctx->view, c);
                    const double d = c[2] - split_g[0] * c[0] - split_g[1] * c[1];
                    z_lo = std::fmin(z_lo, d);
                    z_hi = std::fmax(z_hi, d);
                }
            }
            if (z_hi > z_lo && std::isfinite(z_hi - z_lo)) {
                split_k = std::min(want, c5::kMaxSlabs);
                split_w[0] = -DBL_MAX;
                split_w[split_k] = DBL_MAX;
                // (a hair off the k / K-th: structured grids have whole layers of nodes - and faces - at simple fractions of
                // their depth range, and a face lying IN a cutting plane to rounding makes which of its two cells starts the
                // job above a coin toss per pixel: harmless for the image, but the cell below is then counted by nobody)
                for (int k = 1; k < split_k; ++k) split_w[k] = z_lo + (z_hi - z_lo) * ((static_cast<double>(k) + 0.0309016994) / split_k);
            }
        }
    }
    if (split_k > 1) {
        const int64_t tiles = c5::walk_tiles(im);
        if (fs.split_k != split_k || fs.split_px != padded || fs.split_tiles != tiles || fs.split_cells != g.n_cells) {
            // (re)lay out: plane cells [K - 1][pixels], the sharded list of cells that straddle a plane (room for every
            // cell at every plane: a shard can never overflow), two halves of shard counters (build_records fills one while
            // the other, cleared by the plane raster of the frame before, waits for the next frame), partial results
            // [K][tiles * 64] x (tau, tauc, b: fp64; segments: u32), arrivals [tiles]
            const int64_t waves = (g.n_cells + 63) / 64;
            const int64_t cap = ((waves + c5::kStraddleShards - 1) / c5::kStraddleShards) * 64 * (split_k - 1);
            C5_HIP(ctx, hipStreamSynchronize(s));
            C5_HIP(ctx, fs.plane_cell.ensure(static_cast<size_t>(split_k - 1) * padded * sizeof(uint32_t)));
            C5_HIP(ctx, fs.straddle.ensure(static_cast<size_t>(c5::kStraddleShards) * cap * sizeof(uint32_t)));
            C5_HIP(ctx, fs.straddle_count.ensure(2 * c5::kStraddleShards * c5::kStraddleCounterStride * sizeof(uint32_t)));
            C5_HIP(ctx, fs.partials.ensure(static_cast<size_t>(split_k) * tiles * 64 * 28));
            C5_HIP(ctx, fs.arrivals.ensure(static_cast<size_t>(tiles) * sizeof(uint32_t)));
            C5_HIP(ctx, hipMemsetAsync(fs.plane_cell.ptr, 0, fs.plane_cell.bytes, s));
            C5_HIP(ctx, hipMemsetAsync(fs.straddle_count.ptr, 0, fs.straddle_count.bytes, s));
            C5_HIP(ctx, hipMemsetAsync(fs.arrivals.ptr, 0, fs.arrivals.bytes, s));
            fs.split_k = split_k;
            fs.split_px = padded;
            fs.split_tiles = tiles;
            fs.split_cells = g.n_cells;
            fs.split_seq = 0;
            fs.setup_epoch = 0;  // whatever plane cells the slot held are gone
        }
    }
    // "view_cache" (the persistent device grid of a -D sweep: only the donor turns, main.cpp:112-116): transformed
    // vertices, records and entry lists depend on the grid, the image, the view, the alpha limit and the order - a frame
    // that has all of them in common with the frame before reuses them.  The walk normally hands the entry heads back
    // cleared, so it takes TWO frames with the same view in a row before there is something to reuse: the second one
    // builds everything once more and tells its walk to leave the heads alone; the third and later ones skip the
    // per-view setup.  A sweep whose view changes every frame never pays for any of this.
    const bool cacheable = ctx->view_cache && !ctx->pipeline && !bin_sort && !ctx->fuse_setup && !ctx->overlap_setup && g.n_cells > 0;
    const bool same_view = cacheable && fs.setup_epoch == ctx->setup_epoch && same_rotations(fs.setup_view, ctx->view) &&
                           fs.setup_limit == ctx->alpha_limit && fs.setup_order == ctx->order && fs.setup_split == split_k &&
                           std::memcmp(fs.setup_w, split_w, sizeof split_w) == 0 && std::memcmp(fs.setup_g, split_g, sizeof split_g) == 0;
    const bool reuse = same_view && fs.setup_kept;
    fs.setup_reused = reuse;
    fs.setup_epoch = cacheable ? ctx->setup_epoch : 0;
    fs.setup_view = ctx->view;
    fs.setup_limit = ctx->alpha_limit;
    fs.setup_order = ctx->order;
    fs.setup_kept = same_view;  // (this frame's walk leaves the heads in place)
    fs.setup_split = split_k;
    std::memcpy(fs.setup_w, split_w, sizeof split_w);
    std::memcpy(fs.setup_g, split_g, sizeof split_g);
    c5::SplitParams sp{};
    if (split_k > 1) {
        if (!reuse) {
            fs.split_seq += 1;  // a new set of plane cells: a new stamp (the words of the 15 frames before it stay behind, invalid)
            if (fs.split_seq % 15 == 0) C5_HIP(ctx, hipMemsetAsync(fs.plane_cell.ptr, 0, fs.plane_cell.bytes, s));  // stamps come round
        }
        const int64_t tiles = fs.split_tiles;
        const int64_t waves = (g.n_cells + 63) / 64;
        sp.n_slabs = split_k;
        sp.stamp = static_cast<uint32_t>(fs.split_seq % 15) + 1u;
        std::memcpy(sp.w, split_w, sizeof split_w);
        sp.gx = split_g[0];
        sp.gy = split_g[1];
        sp.plane_cell = fs.plane_cell.as<uint32_t>();
        sp.plane_stride = padded;
        sp.straddle = fs.straddle.as<uint32_t>();
        const size_t half = static_cast<size_t>(c5::kStraddleShards) * c5::kStraddleCounterStride;
        sp.straddle_count = fs.straddle_count.as<uint32_t>() + (fs.split_seq & 1u) * half;
        sp.straddle_count_next = fs.straddle_count.as<uint32_t>() + ((fs.split_seq + 1u) & 1u) * half;
        sp.straddle_capacity = static_cast<uint32_t>(((waves + c5::kStraddleShards - 1) / c5::kStraddleShards) * 64 * (split_k - 1));
        sp.part_stride = tiles * 64;
        char* const base = static_cast<char*>(fs.partials.ptr);
        const size_t n = static_cast<size_t>(split_k) * sp.part_stride;
        sp.part_tau = reinterpret_cast<double*>(base);
        sp.part_tauc = reinterpret_cast<double*>(base + n * 8);
        sp.part_b = reinterpret_cast<double*>(base + n * 16);
        sp.part_nseg = reinterpret_cast<uint32_t*>(base + n * 24);
        sp.arrivals = fs.arrivals.as<uint32_t>();
    }
    g.split = sp;
    if (reuse) {
        c5::launch_clear_walk_counters(s, counters, sb, n_sb, fs.raster_counters);
    } else {
        c5::launch_transform_soa(s, g.px, g.py, g.pz, g.vx, g.vy, g.vz, g.n_pts, ctx->view, counters, sb, n_sb);
        fs.raster_counters = counters;  // (this frame's raster adds its pool demand / overflow here)
    }
    C5_HIP(ctx, mark(1, s));
    if (bin_sort) return enqueue_bin_sort(ctx, fs, g, slot_id, out_dev, s, main_s, timed, counters);
    // (a1, a10, a13 constants) per-cell records on the setup stream; the boundary entry lists and the
    // solid mask need only the transformed vertices, so they run beside it on a side stream (they are
    // small, latency-bound launches).  With stage timing on, everything stays in one stream so that
    // the per-stage times mean something.
    const bool side = ctx->overlap_setup && !timed;
    hipStream_t e = side ? ctx->side_stream : s;
    if (side) {
        C5_HIP(ctx, hipEventRecord(ctx->fork_ev, s));
        C5_HIP(ctx, hipStreamWaitEvent(e, ctx->fork_ev, 0));
    }
    // the frame's uniform entry-key slack (walk_common.hpp: entry_key_slack): a fraction of the GRID's size — not of
    // the image domain's: a slack larger than a whole ray would let a pixel that two boundary faces both claim (its
    // centre exactly on their common edge) walk the same cells twice — plus the rounding of an absolute depth
    const double key_slack = !ctx->entry_key ? -1.0 : c5::kEntryKeySlack * ctx->grid_diagonal + 0x1p-40 * ctx->coord_max;
    if (!(ctx->fuse_setup && !side && g.n_cells > 0) && !reuse) {
        if (ctx->entry_records && !ctx->fuse_setup && !side && g.n_bfaces > 0) {  // (the raster must run BEHIND build_records)
            // build_records leaves a 96-byte record per boundary face a ray can enter through (entry_raster_rec)
            if (fs.bfrec.bytes < static_cast<size_t>(g.n_bfaces) * sizeof(c5::BFaceRecord)) {
                C5_HIP(ctx, fs.bfrec.ensure(static_cast<size_t>(g.n_bfaces) * sizeof(c5::BFaceRecord)));
                C5_HIP(ctx, hipMemsetAsync(fs.bfrec.ptr, 0, fs.bfrec.bytes, s));
                fs.bf_seq = 0;
            }
            fs.bf_seq += 1;
            if (fs.bf_seq == 0) {  // (4 billion frames on: no stale record may look current)
                C5_HIP(ctx, hipMemsetAsync(fs.bfrec.ptr, 0, fs.bfrec.bytes, s));
                fs.bf_seq = 1;
            }
            g.bfrec = fs.bfrec.as<c5::BFaceRecord>();
            g.bf_seq = fs.bf_seq;
            g.bf_want_upper = ctx->order != 0;
            g.bf_key_slack = key_slack;
        }
        // (a cell's optics ride in its record since round 3 — one line per cell and step — and are rewritten with it)
        c5::launch_build_records(s, g, ctx->alpha_limit, ctx->order);
        c5::launch_plane_raster(s, g, ctx->xtab.as<double>(), ctx->ytab.as<double>(), im);  // ("depth_split"; nothing otherwise)
    }
    const bool fused = ctx->fuse_setup && !side && g.n_cells > 0;
    if (!fused) C5_HIP(ctx, mark(2, s));
    // boundary entries: one raster pass (per-pixel count + first entry + overflow chain)
    if (!fs.head_clean && !reuse) C5_HIP(ctx, hipMemsetAsync(fs.head.ptr, 0, static_cast<size_t>(padded) * sizeof(c5::EntryHead), e));
    fs.head_clean = false;
    // ... and a mark on every 8x8 tile that holds one ("tile_flags"; the default tile shape's tiling; not with "fuse_setup")
    uint32_t* tile_flag = nullptr;
    if (!reuse) {
        fs.flags_valid = false;
        if (ctx->tile_flags && ctx->tile_shape == 3 && ctx->lds_stage && !fused && g.n_cells > 0 && g.n_bfaces > 0) {
            const int64_t tiles = c5::walk_tiles(im);
            if (fs.flag_tiles != tiles || !fs.tile_flag.ptr) {
                C5_HIP(ctx, fs.tile_flag.ensure(static_cast<size_t>(tiles) * sizeof(uint32_t)));
                C5_HIP(ctx, hipMemsetAsync(fs.tile_flag.ptr, 0, static_cast<size_t>(tiles) * sizeof(uint32_t), e));
                fs.flag_tiles = tiles;
                fs.flag_seq = 0;
            }
            fs.flag_seq += 1;
            if (fs.flag_seq == 0) {  // (4 billion raster runs on: no old mark may look current)
                C5_HIP(ctx, hipMemsetAsync(fs.tile_flag.ptr, 0, static_cast<size_t>(tiles) * sizeof(uint32_t), e));
                fs.flag_seq = 1;
            }
            tile_flag = fs.tile_flag.as<uint32_t>();
            fs.flags_valid = true;
        }
    }
    if (fused) {
        // records and entry lists as ONE launch of interleaved workgroups ("fuse_setup"; ms_records then holds the
        // time of both and ms_entries is zero)
        c5::launch_setup_fused(s, g, ctx->alpha_limit, ctx->order, ctx->xtab.as<double>(), ctx->ytab.as<double>(), im,
                               fs.head.as<c5::EntryHead>(), fs.first.as<c5::Entry>(), fs.pool.as<c5::Entry>(), fs.entry_capacity,
                               counters, ctx->sticky.as<unsigned>(), ctx->order != 0, key_slack);
        C5_HIP(ctx, mark(2, s));
    } else if (g.n_cells > 0 && !reuse) {
        c5::launch_entry_lists(e, g, ctx->xtab.as<double>(), ctx->ytab.as<double>(), im, fs.head.as<c5::EntryHead>(),
                               fs.first.as<c5::Entry>(), fs.pool.as<c5::Entry>(), fs.entry_capacity, counters,
                               ctx->sticky.as<unsigned>(), ctx->order != 0, key_slack, tile_flag, fs.flag_seq);
    }
    C5_HIP(ctx, mark(3, e));
    // (a9) solids
    c5::SolidTable table{};
    bool any_solid = false;
    rc = enqueue_solids(ctx, fs, slot_id, e, table, any_solid);
    if (rc) return rc;
    C5_HIP(ctx, mark(4, e));
    if (side) {
        C5_HIP(ctx, hipEventRecord(ctx->join_ev, e));
        C5_HIP(ctx, hipStreamWaitEvent(s, ctx->join_ev, 0));
    }

    // (a11-a14) walk on the main stream, after this slot's setup
    c5::WalkParams wp{};
    wp.xrec = g.xrec;
    wp.entry_head = fs.head.as<c5::EntryHead>();
    wp.entry_first = fs.first.as<c5::Entry>();
    wp.entry_pool = fs.pool.as<c5::Entry>();
    wp.pool_capacity = fs.entry_capacity;
    wp.key_slack = key_slack > 0.0 ? key_slack : 0.0;
    wp.mask = any_solid ? fs.mask.as<uint32_t>() : nullptr;
    wp.solids = table;
    wp.Xtab = ctx->xtab.as<double>();
    wp.Ytab = ctx->ytab.as<double>();
    wp.out = out_dev;
    wp.im = im;
    wp.t_cutoff = ctx->t_cutoff;
    wp.max_steps = static_cast<uint32_t>(ctx->n_cells + 64);
    wp.xcd_mode = ctx->xcd_mode;
    wp.lds_pad = ctx->lds_pad;
    wp.stage_slots = ctx->stage_slots ? ctx->stage_slots : (ctx->rays_per_cell > 0.0 && ctx->rays_per_cell < 120.0 ? 21 : 14);
    wp.band_rows = ctx->band_rows;
    wp.order = ctx->order;
    // walk_composite_lds addresses the records by 32-bit byte offsets: n_cells * 128 must fit
    wp.lds_stage = (ctx->lds_stage && ctx->n_cells < (int64_t{1} << 25)) ? ctx->lds_stage : 0;
    wp.counters = counters;
    // (a pixel a solid covers is written by its wavefront whether or not the grid is there: with solids every tile is read)
    wp.tile_flag = (fs.flags_valid && !any_solid && ctx->tile_flags) ? fs.tile_flag.as<uint32_t>() : nullptr;
    wp.tile_stamp = fs.flag_seq;
    {   // every exp argument of this grid within (-1/8, 0]?  alpha_c <= min(limit, largest alpha), chord <= longest edge
        double a_max = std::fmin(ctx->alpha_top, ctx->alpha_limit);
        if (!(a_max >= 0.0)) a_max = ctx->alpha_top;  // (a NaN limit clamps nothing: line.cpp:216-218)
        wp.small_exp_only = (a_max * ctx->edge_max < 0.125) ? 1 : 0;
    }
    wp.keep_entries = fs.setup_kept ? 1 : 0;
    wp.split = sp;
    wp.row_cost = nullptr;
    wp.sb_cost = sb;
    wp.n_sb_rows = 0;
    if (sb && ctx->sb_order_key == sb_key && ctx->sb_order_n == n_sb) {  // an order worked out for this very tiling
        wp.n_sb_rows = n_sb;
        std::memcpy(wp.sb_order, ctx->sb_order, sizeof wp.sb_order);
    }
    wp.sticky = ctx->sticky.as<unsigned>();
    if (ctx->row_costs && im.n_local_rows > 0) {
        wp.row_cost = fs.row_cost.as<uint32_t>();
        C5_HIP(ctx, hipMemsetAsync(fs.row_cost.ptr, 0, static_cast<size_t>(im.n_local_rows) * sizeof(uint32_t), s));
        ctx->row_costs_collected = true;
        ctx->row_cost_slot = slot_id;  // ("pipeline" alternates the slots: the costs stay in the slot of the frame that counted them)
    }
    if (ctx->pipeline) {
        C5_HIP(ctx, hipEventRecord(fs.setup_done, s));
        C5_HIP(ctx, hipStreamWaitEvent(main_s, fs.setup_done, 0));
    }

    int ev_slot = -1;
    // ("walk_timing" N: events around every N-th launch - two events cost 6 us of a 0.53-ms frame when frames follow one another)
    if (ctx->walk_timing && (ctx->walk_seq++ % static_cast<unsigned>(ctx->walk_timing)) == 0u) {
        if (ctx->walk_used == kWalkEventPool) {  // fold the pool before reusing it
            for (int k = 0; k < kWalkEventPool; ++k) {
                C5_HIP(ctx, hipEventSynchronize(ctx->walk_b[k]));
                float ms = 0.f;
                C5_HIP(ctx, hipEventElapsedTime(&ms, ctx->walk_a[k], ctx->walk_b[k]));
                ctx->walk_ms_sum += ms;
            }
            ctx->walk_launches += kWalkEventPool;
            ctx->walk_used = 0;
        }
        ev_slot = ctx->walk_used++;
        C5_HIP(ctx, hipEventRecord(ctx->walk_a[ev_slot], main_s));
    }
    c5::launch_walk(main_s, wp, ctx->tile_shape);
    fs.head_clean = !wp.keep_entries;  // stream order: every pixel's head is zero again once the walk has run
    if (ev_slot >= 0) C5_HIP(ctx, hipEventRecord(ctx->walk_b[ev_slot], main_s));
    C5_HIP(ctx, mark(5, main_s));
    C5_HIP(ctx, hipGetLastError());
    // the statistics of the last frame and the sticky failure words are fetched when somebody waits for
    // the stream (wait_and_collect), not once per frame: two API calls and two small copies less per frame
    ctx->counters_on_host = false;
    if (ctx->pipeline) {
        C5_HIP(ctx, hipEventRecord(fs.walk_done, main_s));
        fs.walk_recorded = true;
    }
    ctx->last_slot = slot_id;
    ctx->last_counters = counters;
    ctx->frame_index += 1;
    ctx->frame_pending = true;
    ctx->frame_timed = timed;
    return C5_OK;
}

int finish_frame(c5_context* ctx);

// Wait for the context's stream and collect the last frame's outcome.
int wait_and_collect(c5_context* ctx) {
    if (ctx->frame_pending && !ctx->counters_on_host) {
        FrameSlot& fs = ctx->slots[ctx->last_slot];
        C5_HIP(ctx, hipMemcpyAsync(fs.host_counters, ctx->last_counters ? ctx->last_counters : fs.counters.ptr, kCountersBytes, hipMemcpyDeviceToHost, ctx->stream));
        C5_HIP(ctx, hipMemcpyAsync(ctx->host_sticky, ctx->sticky.ptr, kStickyWords * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        if (fs.sb_key >= 0 && fs.sb.ptr && ctx->host_sb)
            C5_HIP(ctx, hipMemcpyAsync(ctx->host_sb, fs.sb.ptr, c5::kMaxSbRows * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        ctx->counters_on_host = true;
    }
    C5_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return finish_frame(ctx);
}

// After the stream drained: collect counters/timings; grow the entry buffer if it overflowed.
int finish_frame(c5_context* ctx) {
    if (!ctx->frame_pending) return C5_OK;
    ctx->frame_pending = false;
    FrameSlot& fs = ctx->slots[ctx->last_slot];
    struct {  // the shards' sums, the packed pairs taken apart (device_types.hpp: FrameCounters)
        unsigned long long segments = 0, steps = 0, covered = 0, solid_pixels = 0, entries = 0, ray_tiles = 0, exit_max_key = 0, entry_min_key = 0;
        unsigned walk_overflow = 0, entry_overflow = 0, odd_pixels = 0, pool_used = 0, seg_max = 0;
    } hc;
    for (int k = 0; k < c5::kCounterShards; ++k) {
        const c5::FrameCounters& p = fs.host_counters[k];
        hc.segments += p.seg_tiles & c5::kCounterLowMask;
        hc.ray_tiles += p.seg_tiles >> c5::kCounterHighShift;
        hc.steps += p.steps_cov & c5::kCounterLowMask;
        hc.covered += p.steps_cov >> c5::kCounterHighShift;
        hc.entries += p.ent_solid & c5::kCounterLowMask;
        hc.solid_pixels += p.ent_solid >> c5::kCounterHighShift;
        hc.walk_overflow += p.walk_overflow;
        hc.entry_overflow += p.entry_overflow;
        hc.odd_pixels += p.odd_pixels;
        hc.pool_used += p.pool_used;
        hc.seg_max = std::max(hc.seg_max, p.seg_max);
        hc.exit_max_key = std::max(hc.exit_max_key, p.exit_max_key);
        hc.entry_min_key = std::max(hc.entry_min_key, p.entry_min_key);
    }
    c5_stats& st = ctx->last;
    // The order the next frames' rows of super-blocks start in.  A frame with fewer wavefronts of rays than about two
    // rounds of the GPU's wavefront slots lasts as long as its longest wavefronts plus the time the dispatcher takes
    // to reach them behind thousands of empty or short tiles, and ends on whatever started last: such frames start
    // the rows with the LONGEST RAYS first (by this frame's longest ray per row - of one tile per super-block - in eight
    // classes of the longest of all, so that rows of about the same length keep their image order).  Round 4 (profiles/
    // experiments.md): the key used to be the row's SUM of segments, which sent a cut-off row of full-length rays at the
    // lower edge of a share to the very end (the upper half of the C3 frame: 0.318 -> 0.347 ms with the order, 0.301 with
    // this one; an eighth of the 4800x3600 frame at the image's edge 0.307 -> 0.247).  Larger frames (several rounds of
    // wavefronts) lose a little (C3 frame 0.534 -> 0.542 ms) and stay in image order; so do frames of a few hundred
    // wavefronts (C2 ball at 600x450: 0.106 -> 0.112).
    ctx->sb_order_key = -1;
    ctx->sb_order_n = 0;
    constexpr unsigned long long kSmallFrameRays = 2ull * 256 * 32 * 64;  // two rounds of 8 wavefronts per SIMD
    constexpr unsigned long long kTinyFrameRays = 100000;                  // ~1 500 wavefronts
    if (fs.sb_key >= 0 && ctx->host_sb && fs.sb_n > 1 && fs.sb_n <= c5::kMaxSbRows &&
        ((hc.covered >= kTinyFrameRays && hc.covered < kSmallFrameRays) || (hc.covered > 0 && ctx->cost_order == 2))) {
        const int n = fs.sb_n;
        uint32_t top = 0;
        for (int j = 0; j < n; ++j) top = std::max(top, ctx->host_sb[j]);
        const uint32_t unit = top / 8u + 1u;
        int order[c5::kMaxSbRows];
        for (int j = 0; j < n; ++j) order[j] = j;
        std::stable_sort(order, order + n, [&](int a, int b) { return ctx->host_sb[a] / unit > ctx->host_sb[b] / unit; });
        for (int j = 0; j < n; ++j) ctx->sb_order[j] = static_cast<uint8_t>(order[j]);
        ctx->sb_order_key = fs.sb_key;
        ctx->sb_order_n = n;
    }
    st.segments = static_cast<int64_t>(hc.segments);
    // How coarse the pixels are against the cells decides how many distinct cells an 8x8 tile meets per step, and
    // with it how many staging slots the next frame's walk gets (walk_kernels.hip: 14 or 21): rays per cell of the
    // WHOLE frame, this context's share scaled up by the number of row shards.
    if (ctx->n_cells > 0 && hc.segments > 0)
        ctx->rays_per_cell = static_cast<double>(hc.segments) * static_cast<double>(ctx->im.world > 0 ? ctx->im.world : 1) *
                             (static_cast<double>(ctx->im.res_y) / static_cast<double>(ctx->im.row_count > 0 ? ctx->im.row_count : ctx->im.res_y)) /
                             static_cast<double>(ctx->n_cells);
    {   // "depth_split" 0: how many slabs the next frames' rays are cut into.  K jobs per tile of a K-th of a ray's steps each:
        // worth it while the jobs do not fill the wavefront slots (a frame of one round lasts as long as ONE ray, however
        // few rays it has) and the rays are long enough to be worth cutting.
        // Measured (profiles/r04_split_probe.md): K jobs per tile pay while K x (tiles with rays) still fit the slots in ONE
        // round — the 124-row share of the C3 frame that one of 8 GPUs renders: walk 0.19 -> 0.13 (2 slabs); 4 slabs, 8 640
        // jobs on 7 168 slots, are two rounds and no faster than 2 — and a slab is worth its plane raster and its jobs'
        // start and end (~40 segments per ray and slab: the C2 ball's 70-segment rays are left whole).
        // What a frame lasts is set by its LONGEST rays (seg_max), and they are cut evenly only if the planes divide THEIR
        // depth range: the next frame's planes go between the shallowest entry and the deepest exit this frame's rays had
        // (one GPU's rows of a frame see a part of the grid's depth range only; no earlier frame: the grid's bounding box).
        int k = 1;
        if (hc.covered > 0 && hc.segments > 0 && hc.ray_tiles > 0) {
            // jobs that really walk: a tile's rays span about K x (their length / the longest ray's) slabs, + 1/2 for the
            // plane they straddle; the other jobs of the tile find nothing to do and leave their slot at once
            const double slots = 0.98 * 256.0 * 4.0 * 7.0;  // the split walk runs 7 wavefronts per SIMD
            const double mean_over_max = std::fmin(1.0, static_cast<double>(hc.segments) / static_cast<double>(hc.covered) / static_cast<double>(hc.seg_max ? hc.seg_max : 1u));
            k = 1;
            for (int t = 2; t <= 4; ++t)
                if (static_cast<double>(hc.ray_tiles) * (t * mean_over_max + 0.5) <= slots) k = t;
            k = std::min(k, static_cast<int>(static_cast<double>(hc.seg_max) / 56.0));
            k = std::max(1, k);
        }
        {   // the planes through the sampled rays' entries and ends: z ~ a + b x + c y by least squares, each; then the mean
            // tilt, and what is left of either plane's depth at the origin once that tilt is taken out of its sample
            const c5::DepthSamples& smp = *reinterpret_cast<const c5::DepthSamples*>(fs.host_counters + c5::kCounterShards);
            struct { double entry[9], exit_[9]; } fit{};  // n, Sx, Sy, Sxx, Sxy, Syy, Sz, Sxz, Syz
            {
                const c5::ImageParams& im = ctx->im;  // (c5_set_image drains the ring before it changes it)
                const int n_slots = std::min<int64_t>(c5::kFitSlots, static_cast<int64_t>(im.fit_cols) * ((im.res_y >> im.fit_shift) + 1));
                const int mid = ((1 << im.fit_shift) - 1) >> 1;
                for (int slot = 0; slot < n_slots; ++slot) {
                    if (smp.entry_key[slot] == 0ull || smp.exit_key[slot] == 0ull) continue;
                    const double x = im.x_min + im.step_x * (((slot % im.fit_cols) << im.fit_shift) + mid);
                    const double y = im.y_min + im.step_y * (((slot / im.fit_cols) << im.fit_shift) + mid);
                    const double z[2] = {c5::depth_of_key(smp.entry_key[slot]), c5::depth_of_key(smp.exit_key[slot])};
                    double* const sums[2] = {fit.entry, fit.exit_};
                    for (int t = 0; t < 2; ++t) {
                        const double v[9] = {1.0, x, y, x * x, x * y, y * y, z[t], x * z[t], y * z[t]};
                        for (int j = 0; j < 9; ++j) sums[t][j] += v[j];
                    }
                }
            }
            auto solve = [](const double* m, double out[3]) {  // normal equations, Cramer (3 x 3, well scaled: x, y ~ 1)
                const double n = m[0], sx = m[1], sy = m[2], sxx = m[3], sxy = m[4], syy = m[5], sz = m[6], sxz = m[7], syz = m[8];
                const double det = n * (sxx * syy - sxy * sxy) - sx * (sx * syy - sxy * sy) + sy * (sx * sxy - sxx * sy);
                if (!(std::fabs(det) > 1e-12 * std::fabs(n * sxx * syy) && n >= 12.0)) return false;
                out[0] = (sz * (sxx * syy - sxy * sxy) - sx * (sxz * syy - sxy * syz) + sy * (sxz * sxy - sxx * syz)) / det;
                out[1] = (n * (sxz * syy - syz * sxy) - sz * (sx * syy - sxy * sy) + sy * (sx * syz - sxz * sy)) / det;
                out[2] = (n * (sxx * syz - sxy * sxz) - sx * (sx * syz - sxz * sy) + sz * (sx * sxy - sxx * sy)) / det;
                return std::isfinite(out[0]) && std::isfinite(out[1]) && std::isfinite(out[2]);
            };
            double pe[3], px[3];
            ctx->fit_known = false;
            if (solve(fit.entry, pe) && solve(fit.exit_, px)) {
                const double gx = 0.5 * (pe[1] + px[1]), gy = 0.5 * (pe[2] + px[2]);
                const double e0 = (fit.entry[6] - gx * fit.entry[1] - gy * fit.entry[2]) / fit.entry[0];
                const double x0 = (fit.exit_[6] - gx * fit.exit_[1] - gy * fit.exit_[2]) / fit.exit_[0];
                if (x0 > e0 && std::fabs(gx) < 64.0 && std::fabs(gy) < 64.0) {
                    ctx->fit_known = true;
                    ctx->fit_gx = gx, ctx->fit_gy = gy, ctx->fit_entry0 = e0, ctx->fit_exit0 = x0;
                }
            }
        }
        ctx->ray_depth_known = hc.exit_max_key != 0 && hc.entry_min_key != 0;
        if (ctx->ray_depth_known) {
            ctx->ray_depth_lo = -c5::depth_of_key(hc.entry_min_key);
            ctx->ray_depth_hi = c5::depth_of_key(hc.exit_max_key);
            ctx->ray_depth_known = ctx->ray_depth_hi > ctx->ray_depth_lo;
        }
        if (ctx->algorithm == 0 && ctx->grid_conforming && !ctx->overlap_seen) ctx->split_auto_k = k;
    }
    st.covered_pixels = static_cast<int64_t>(hc.covered);
    st.solid_pixels = static_cast<int64_t>(hc.solid_pixels);
    st.entries = static_cast<int64_t>(hc.entries);
    st.boundary_faces = ctx->n_bfaces;
    st.steps = static_cast<int64_t>(hc.steps);
    st.walk_overflow = static_cast<int32_t>(hc.walk_overflow);
    st.entry_overflow = hc.entry_overflow > 0 ? 1 : 0;  // THIS frame; the sticky word below covers every frame in flight
    st.odd_pixels = static_cast<int64_t>(hc.odd_pixels);
    if (ctx->frame_timed) {
        float* dst[5] = {&st.ms_transform, &st.ms_records, &st.ms_entries, &st.ms_solids, &st.ms_walk};
        for (int k = 0; k < 5; ++k) C5_HIP(ctx, hipEventElapsedTime(dst[k], fs.ev[k], fs.ev[k + 1]));
        C5_HIP(ctx, hipEventElapsedTime(&st.ms_total, fs.ev[0], fs.ev[5]));
        // a frame that reused the per-view data of the frames before it ("view_cache") ran none of the three: exactly 0
        if (fs.setup_reused) st.ms_transform = st.ms_records = st.ms_entries = 0.0f;
    }
    // overflow entries this frame needed: handed out (a shard hands out min(asked, its part)) + refused
    int64_t handed = 0;
    for (int k = 0; k < c5::kCounterShards; ++k) {
        const int64_t part = (static_cast<int64_t>(k + 1) * fs.entry_capacity) / c5::kCounterShards -
                             (static_cast<int64_t>(k) * fs.entry_capacity) / c5::kCounterShards;
        handed += std::min<int64_t>(static_cast<int64_t>(fs.host_counters[k].pool_used), part);
    }
    st.pool_entries = handed + static_cast<int64_t>(hc.entry_overflow);
    st.pool_capacity = fs.entry_capacity;
    // failures of ANY frame since the last look (several frames may have been in flight)
    const int64_t refused = static_cast<int64_t>(ctx->host_sticky[0]);  // entries that found no slot, all those frames
    const unsigned lost_rays = ctx->host_sticky[1];
    const unsigned overlap_rays = ctx->host_sticky[2];
    const bool too_small = refused > 0;
    if (too_small || lost_rays || overlap_rays) {
        int rc = drain(ctx);
        if (rc) return rc;
        // Frames delivered to host memory that are still outstanding were all enqueued before this moment, i.e.
        // rendered with the buffers that were too small: their waits must say so whatever their status snapshots
        // read (a snapshot is copied on the copy stream and may run after the words are cleared here) — also when
        // it is c5_get_stats / c5_get_row_costs / c5_synchronize, not c5_render_host_wait, that notices first.
        if (too_small || overlap_rays) ctx->hr_retry_left = ctx->hr_count;
        if (ctx->copy_stream && ctx->hr_count) C5_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
        // cleared IN the stream the kernels that add to it run on, and waited for: nothing rests on how the NULL stream
        // is ordered against this context's non-blocking streams
        C5_HIP(ctx, hipMemsetAsync(ctx->sticky.ptr, 0, kStickyWords * sizeof(unsigned), ctx->stream));
        C5_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < kStickyWords; ++k) ctx->host_sticky[k] = 0;
    }
    // Keep the pool at twice the demand just seen (plus a margin): the demand is a smooth function of the
    // view, so in a sweep the pool grows ahead of it, between frames, without a frame ever being lost.
    int64_t want = 0;
    if (too_small) want = 2 * (std::max(st.pool_entries, fs.entry_capacity) + refused) + 8192;
    else if (2 * st.pool_entries + 4096 > fs.entry_capacity) want = 2 * st.pool_entries + 8192;
    if (want > static_cast<int64_t>(16777214) * 64) want = static_cast<int64_t>(16777214) * 64;  // slot + 1 must fit 31 bits
    if (want > fs.entry_capacity) {
        int rc = drain(ctx);
        if (rc) return rc;
        for (int k = 0; k < (ctx->pipeline ? kFrameSlots : 1); ++k) {
            FrameSlot& o = ctx->slots[k];
            if (o.entry_capacity >= want) continue;
            o.entry_capacity = want;
            C5_HIP(ctx, o.pool.ensure(static_cast<size_t>(want) * sizeof(c5::Entry)));
            ++ctx->setup_epoch;  // (the entry lists lived in the old pool)
        }
    }
    if (too_small) {
        return fail(ctx, C5_RETRY,
                    "%lld boundary entries found no room in the overflow pool (now %lld records): every frame "
                    "since the last c5_synchronize is incomplete, render again",
                    static_cast<long long>(refused), static_cast<long long>(ctx->slots[0].entry_capacity));
    }
    if (lost_rays)
        return fail(ctx, C5_ERR_WALK, "%u rays exceeded the walk step bound (malformed grid?)", lost_rays);
    if (overlap_rays) {
        // Cells of two components that share no face interpenetrate (walk_common.hpp: next_entry): the reference bins and
        // sorts such a soup (plane.cpp:184-192, line.cpp:138), a walk cannot render it.  From here on this grid goes
        // through bin_sort_resolve, like a grid with a face in more than two cells (c5_upload_grid).
        ctx->overlap_seen = true;
        ++ctx->setup_epoch;
        return fail(ctx, C5_RETRY,
                    "%u rays met a boundary entry inside a stretch of cells they had walked: components of the grid interpenetrate; "
                    "every frame since the last c5_synchronize is wrong, render again (bin_sort_resolve from now on)", overlap_rays);
    }
    return C5_OK;
}

}  // namespace

extern "C" {

int c5_abi_version(void) { return C5_ABI_VERSION; }

int c5_device_count(int* count) {
    if (!count) return fail(nullptr, C5_ERR_INVALID, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(nullptr, C5_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return C5_OK;
}

int c5_create(int device_ordinal, c5_context** out_ctx) {
    if (!out_ctx) return fail(nullptr, C5_ERR_INVALID, "null out_ctx");
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, C5_ERR_NO_DEVICE, "no HIP device available (%s)",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device_ordinal < 0 || device_ordinal >= n)
        return fail(nullptr, C5_ERR_INVALID, "device ordinal %d out of range [0, %d)", device_ordinal, n);
    c5_context* ctx = new (std::nothrow) c5_context();
    if (!ctx) return fail(nullptr, C5_ERR_INVALID, "out of host memory");
    ctx->device = device_ordinal;
    auto bail = [&](hipError_t err, const char* what) {
        fail(nullptr, C5_ERR_HIP, "%s: %s", what, hipGetErrorString(err));
        c5_destroy(ctx);
        return C5_ERR_HIP;
    };
    if ((e = hipSetDevice(device_ordinal)) != hipSuccess) return bail(e, "hipSetDevice");
    {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if ((e = hipStreamCreateWithPriority(&ctx->own_stream, hipStreamNonBlocking, hi)) != hipSuccess)
            return bail(e, "hipStreamCreate");
    }
    ctx->stream = ctx->own_stream;
    {   // the setup stream only fills what the walk leaves idle: lowest priority
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if ((e = hipStreamCreateWithPriority(&ctx->aux_stream, hipStreamNonBlocking, lo)) != hipSuccess)
            return bail(e, "hipStreamCreate");
    }
    if ((e = hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking)) != hipSuccess)
        return bail(e, "hipStreamCreate");
    if ((e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking)) != hipSuccess)
        return bail(e, "hipStreamCreate");
    for (c5_context::HostFrame& h : ctx->hring) {
        if ((e = hipEventCreateWithFlags(&h.rendered, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
        if ((e = hipEventCreateWithFlags(&h.copied, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
        if ((e = hipHostMalloc(reinterpret_cast<void**>(&h.status), 4 * sizeof(unsigned), hipHostMallocDefault)) != hipSuccess)
            return bail(e, "hipHostMalloc");
        h.status[0] = h.status[1] = h.status[2] = h.status[3] = 0;
    }
    for (int k = 0; k < 2; ++k)
        if ((e = hipEventCreateWithFlags(&ctx->stage_ev[k], hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&ctx->join_ev, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    for (FrameSlot& fs : ctx->slots) {
        for (auto& ev : fs.ev)
            if ((e = hipEventCreate(&ev)) != hipSuccess) return bail(e, "hipEventCreate");
        if ((e = hipEventCreateWithFlags(&fs.setup_done, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
        if ((e = hipEventCreateWithFlags(&fs.walk_done, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
        if ((e = fs.counters.ensure(kCountersBytes)) != hipSuccess) return bail(e, "hipMalloc");
        if ((e = hipHostMalloc(reinterpret_cast<void**>(&fs.host_counters), kCountersBytes, hipHostMallocDefault)) !=
            hipSuccess)
            return bail(e, "hipHostMalloc");
        std::memset(fs.host_counters, 0, kCountersBytes);
    }
    for (int k = 0; k < kWalkEventPool; ++k) {
        ctx->walk_a[k] = ctx->walk_b[k] = nullptr;
    }
    for (int k = 0; k < kWalkEventPool; ++k) {
        if ((e = hipEventCreate(&ctx->walk_a[k])) != hipSuccess) return bail(e, "hipEventCreate");
        if ((e = hipEventCreate(&ctx->walk_b[k])) != hipSuccess) return bail(e, "hipEventCreate");
    }
    // (a line of its own; cleared in the context's own stream and waited for — see finish_frame)
    if ((e = ctx->sticky.ensure(256)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMemsetAsync(ctx->sticky.ptr, 0, 256, ctx->own_stream)) != hipSuccess) return bail(e, "hipMemsetAsync");
    if ((e = hipStreamSynchronize(ctx->own_stream)) != hipSuccess) return bail(e, "hipStreamSynchronize");
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&ctx->host_sticky), 4 * sizeof(unsigned), hipHostMallocDefault)) != hipSuccess)
        return bail(e, "hipHostMalloc");
    for (int k = 0; k < 4; ++k) ctx->host_sticky[k] = 0;
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&ctx->host_sb), c5::kMaxSbRows * sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess)
        ctx->host_sb = nullptr;  // (the walk then starts its rows in image order)
    else
        std::memset(ctx->host_sb, 0, c5::kMaxSbRows * sizeof(uint32_t));
    ctx->view.n = 0;
    for (Solid& s : ctx->solids) s.rots.n = 0;
    *out_ctx = ctx;
    return C5_OK;
}

void c5_destroy(c5_context* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->aux_stream) (void)hipStreamSynchronize(ctx->aux_stream);
    DeviceBuffer* bufs[] = {&ctx->px, &ctx->py, &ctx->pz, &ctx->cell_vert, &ctx->cell_adj, &ctx->alpha,
                            &ctx->q, &ctx->bface, &ctx->xtab, &ctx->ytab, &ctx->out, &ctx->sticky,
                            &ctx->offs64, &ctx->scratch64, &ctx->segs};
    if (ctx->host_sticky) (void)hipHostFree(ctx->host_sticky);
    if (ctx->host_sb) (void)hipHostFree(ctx->host_sb);
    for (DeviceBuffer* b : bufs) b->release();
    for (FrameSlot& fs : ctx->slots) {
        DeviceBuffer* sb[] = {&fs.vx, &fs.vy, &fs.vz, &fs.rec, &fs.count, &fs.head, &fs.first, &fs.pool,
                              &fs.mask, &fs.counters, &fs.row_cost, &fs.sb, &fs.plane_cell, &fs.straddle, &fs.straddle_count,
                              &fs.partials, &fs.arrivals, &fs.bfrec};
        for (DeviceBuffer* b : sb) b->release();
        if (fs.host_counters) (void)hipHostFree(fs.host_counters);
        for (auto& ev : fs.ev)
            if (ev) (void)hipEventDestroy(ev);
        if (fs.setup_done) (void)hipEventDestroy(fs.setup_done);
        if (fs.walk_done) (void)hipEventDestroy(fs.walk_done);
    }
    for (Solid& so : ctx->solids) {
        so.raw.release();
        so.faces.release();
        for (DeviceBuffer& v : so.view) v.release();
        so.own_mask.release();
    }
    for (int k = 0; k < kWalkEventPool; ++k) {
        if (ctx->walk_a[k]) (void)hipEventDestroy(ctx->walk_a[k]);
        if (ctx->walk_b[k]) (void)hipEventDestroy(ctx->walk_b[k]);
    }
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    for (c5_context::HostFrame& h : ctx->hring) {
        h.img.release();
        h.counters.release();
        if (h.rendered) (void)hipEventDestroy(h.rendered);
        if (h.copied) (void)hipEventDestroy(h.copied);
        if (h.status) (void)hipHostFree(h.status);
    }
    for (int k = 0; k < 2; ++k) {
        if (ctx->stage[k]) (void)hipHostFree(ctx->stage[k]);
        if (ctx->stage_ev[k]) (void)hipEventDestroy(ctx->stage_ev[k]);
    }
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
    if (ctx->join_ev) (void)hipEventDestroy(ctx->join_ev);
    delete ctx;
}

int c5_set_stream(c5_context* ctx, void* hip_stream) {
    if (!ctx) return fail(nullptr, C5_ERR_INVALID, "null context");
    ++ctx->setup_epoch;  // whatever was built per view is stale ("view_cache")
    int rc = bind_device(ctx);
    if (rc) return rc;
    rc = wait_and_collect(ctx);
    ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    ctx->using_caller_stream = hip_stream != nullptr;
    return rc;  // C5_RETRY included: the stream IS switched, but the frames before the switch must be rendered again
}

const char* c5_last_error(const c5_context* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int c5_upload_grid(c5_context* ctx, const double* xyz, int64_t n_pts, const int32_t* cell_vert,
                   int64_t n_cells, const double* alpha, const double* q) {
    if (!ctx) return fail(nullptr, C5_ERR_INVALID, "null context");
    ++ctx->setup_epoch;  // whatever was built per view is stale ("view_cache")
    if (n_pts < 0 || n_cells < 0) return fail(ctx, C5_ERR_INVALID, "negative size");
    if (n_cells > 0 && (!xyz || !cell_vert || !alpha || !q)) return fail(ctx, C5_ERR_INVALID, "null grid array");
    if (n_cells >= static_cast<int64_t>(c5::kNoCell))
        return fail(ctx, C5_ERR_INVALID, "cell count %lld does not fit 28 bits (line.hpp:71-79)",
                    static_cast<long long>(n_cells));
    for (int64_t i = 0; i < 3 * n_pts; ++i)
        if (!std::isfinite(xyz[i])) return fail(ctx, C5_ERR_INVALID, "point %lld has a non-finite coordinate", static_cast<long long>(i / 3));
    int rc = bind_device(ctx);
    if (rc) return rc;

    for (int64_t i = 0; i < 4 * n_cells; ++i)
        if (cell_vert[i] < 0 || cell_vert[i] >= n_pts)
            return fail(ctx, C5_ERR_INVALID, "cell %lld references a point id out of range", static_cast<long long>(i / 4));
    // Coincident points are one point: the reference copies coordinates per cell (object3d_base.cpp:37-42)
    // and never sees ids, so files with per-cell point copies or duplicated seam points must walk like any
    // other grid (without this every face of such a file would be a boundary face).
    std::vector<int32_t> welded;
    {
        std::vector<int32_t> rep;
        if (c5::weld_points(xyz, n_pts, rep) > 0) {
            welded.resize(static_cast<size_t>(4 * n_cells));
            for (int64_t i = 0; i < 4 * n_cells; ++i) welded[static_cast<size_t>(i)] = rep[static_cast<size_t>(cell_vert[i])];
            cell_vert = welded.data();
        }
    }
    // "cell_order" (round 4): the cells are kept in Morton order of their centroids, whatever order the caller has them in
    // - 256 consecutive cells are then a compact lump of the grid, which is what lets build_records drop whole workgroups
    // by a sphere about their cells (GridView::block_sphere) and keeps a boundary face's record near its neighbours'.
    // Nothing of it shows outside: images are bit-equal (a cell's own arithmetic does not know its number), and
    // c5_update_scalars takes its arrays in the caller's order.  Not for grids that go to bin_sort_resolve (below).
    std::vector<int32_t> perm;  // new index -> the caller's
    std::vector<int32_t> ordered;
    const int32_t* caller_cell_vert = cell_vert;
    if (ctx->cell_order && n_cells >= 4096) {
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        std::vector<double> cen(static_cast<size_t>(3 * n_cells));
        for (int64_t c = 0; c < n_cells; ++c)
            for (int k = 0; k < 3; ++k) {
                double m = 0.0;
                for (int a = 0; a < 4; ++a) m += xyz[3 * static_cast<int64_t>(cell_vert[4 * c + a]) + k];
                cen[static_cast<size_t>(3 * c + k)] = m;
                lo[k] = std::fmin(lo[k], m);
                hi[k] = std::fmax(hi[k], m);
            }
        auto spread = [](uint64_t v) {  // 10 bits -> every third bit
            v &= 0x3ffull;
            v = (v | (v << 16)) & 0x30000ffull;
            v = (v | (v << 8)) & 0x300f00full;
            v = (v | (v << 4)) & 0x30c30c3ull;
            v = (v | (v << 2)) & 0x9249249ull;
            return v;
        };
        std::vector<std::pair<uint64_t, int32_t>> keyed(static_cast<size_t>(n_cells));
        for (int64_t c = 0; c < n_cells; ++c) {
            uint64_t key = 0;
            for (int k = 0; k < 3; ++k) {
                const double span = hi[k] - lo[k];
                const double t = span > 0.0 ? (cen[static_cast<size_t>(3 * c + k)] - lo[k]) / span * 1024.0 : 0.0;
                key |= spread(static_cast<uint64_t>(std::fmin(std::fmax(t, 0.0), 1023.0))) << k;
            }
            keyed[static_cast<size_t>(c)] = {key, static_cast<int32_t>(c)};
        }
        std::sort(keyed.begin(), keyed.end());  // (ties by the caller's index: a total order, the same on every rank)
        bool identity = true;
        for (int64_t c = 0; c < n_cells && identity; ++c) identity = keyed[static_cast<size_t>(c)].second == c;
        if (!identity) {
            perm.resize(static_cast<size_t>(n_cells));
            ordered.resize(static_cast<size_t>(4 * n_cells));
            for (int64_t c = 0; c < n_cells; ++c) {
                const int32_t from = keyed[static_cast<size_t>(c)].second;
                perm[static_cast<size_t>(c)] = from;
                for (int a = 0; a < 4; ++a) ordered[static_cast<size_t>(4 * c + a)] = cell_vert[4 * static_cast<int64_t>(from) + a];
            }
            cell_vert = ordered.data();
        }
    }
    std::vector<int32_t> adj;
    std::vector<uint32_t> bfaces;
    std::string err;
    bool conforming = true;
    if (!c5::build_face_adjacency(cell_vert, n_cells, n_pts, adj, bfaces, err)) {
        cell_vert = caller_cell_vert;  // (bin_sort_resolve breaks ties of equal depths by the cells' order: the caller's stays)
        perm.clear();
        if (err.find("range") != std::string::npos) return fail(ctx, C5_ERR_INVALID, "%s", err.c_str());
        // a face shared by more than two cells: no walk possible, the reference's own algorithm will do
        conforming = false;
        adj.assign(static_cast<size_t>(4 * n_cells), -1);
        bfaces.clear();
    }

    // SoA split of the points
    std::vector<double> sx(static_cast<size_t>(n_pts)), sy(static_cast<size_t>(n_pts)), sz(static_cast<size_t>(n_pts));
    for (int64_t i = 0; i < n_pts; ++i) {
        sx[static_cast<size_t>(i)] = xyz[3 * i];
        sy[static_cast<size_t>(i)] = xyz[3 * i + 1];
        sz[static_cast<size_t>(i)] = xyz[3 * i + 2];
    }
    const size_t pb = static_cast<size_t>(n_pts) * sizeof(double);
    const size_t cb = static_cast<size_t>(n_cells);
    C5_HIP(ctx, hipStreamSynchronize(ctx->stream));
    DeviceBuffer* pbufs[] = {&ctx->px, &ctx->py, &ctx->pz};
    for (DeviceBuffer* b : pbufs) C5_HIP(ctx, b->ensure(pb ? pb : 8));
    for (int k = 0; k < (ctx->pipeline ? kFrameSlots : 1); ++k) {
        FrameSlot& fs = ctx->slots[k];
        C5_HIP(ctx, fs.vx.ensure(pb ? pb : 8));
        C5_HIP(ctx, fs.vy.ensure(pb ? pb : 8));
        C5_HIP(ctx, fs.vz.ensure(pb ? pb : 8));
        // one 128-byte record per cell and view (ExitRecord)
        C5_HIP(ctx, fs.rec.ensure(cb * sizeof(c5::ExitRecord) + 256));
        // records of cells outside a context's row band are never rebuilt; keep whatever they hold a
        // valid record (neighbour ids inside the grid) from the start
        C5_HIP(ctx, hipMemset(fs.rec.ptr, 0, fs.rec.bytes));
    }
    C5_HIP(ctx, ctx->cell_vert.ensure(cb * 16 + 16));
    C5_HIP(ctx, ctx->cell_adj.ensure(cb * 16 + 16));
    C5_HIP(ctx, ctx->alpha.ensure(cb * 8 + 8));
    C5_HIP(ctx, ctx->q.ensure(cb * 8 + 8));
    C5_HIP(ctx, ctx->bface.ensure(bfaces.size() * 4 + 4));
    if (n_pts > 0) {
        C5_HIP(ctx, hipMemcpy(ctx->px.ptr, sx.data(), pb, hipMemcpyHostToDevice));
        C5_HIP(ctx, hipMemcpy(ctx->py.ptr, sy.data(), pb, hipMemcpyHostToDevice));
        C5_HIP(ctx, hipMemcpy(ctx->pz.ptr, sz.data(), pb, hipMemcpyHostToDevice));
    }
    if (n_cells > 0) {
        C5_HIP(ctx, hipMemcpy(ctx->cell_vert.ptr, cell_vert, cb * 16, hipMemcpyHostToDevice));
        // the device's copy names every boundary face by its index in the sorted boundary-face list: -(i + 2) where the
        // host API says -1 (build_records leaves the face's record in slot i: device_types.hpp: BFaceRecord)
        std::vector<int32_t> adj_dev(adj);
        for (size_t i = 0; i < bfaces.size(); ++i)
            adj_dev[static_cast<size_t>(bfaces[i] >> 2) * 4 + (bfaces[i] & 3u)] = -static_cast<int32_t>(i) - 2;
        C5_HIP(ctx, hipMemcpy(ctx->cell_adj.ptr, adj_dev.data(), cb * 16, hipMemcpyHostToDevice));
        std::vector<double> a_dev, q_dev;
        const double *a_src = alpha, *q_src = q;
        if (!perm.empty()) {
            a_dev.resize(cb);
            q_dev.resize(cb);
            for (size_t c = 0; c < cb; ++c) a_dev[c] = alpha[perm[c]], q_dev[c] = q[perm[c]];
            a_src = a_dev.data(), q_src = q_dev.data();
        }
        C5_HIP(ctx, hipMemcpy(ctx->alpha.ptr, a_src, cb * 8, hipMemcpyHostToDevice));
        C5_HIP(ctx, hipMemcpy(ctx->q.ptr, q_src, cb * 8, hipMemcpyHostToDevice));
    }
    ctx->cell_perm = std::move(perm);
    if (!bfaces.empty())
        C5_HIP(ctx, hipMemcpy(ctx->bface.ptr, bfaces.data(), bfaces.size() * 4, hipMemcpyHostToDevice));
    {
        double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, top = 0.0;
        for (int64_t i = 0; i < n_pts; ++i)
            for (int k = 0; k < 3; ++k) {
                const double v = xyz[3 * i + k];
                lo[k] = i ? std::fmin(lo[k], v) : v;
                hi[k] = i ? std::fmax(hi[k], v) : v;
                top = std::fmax(top, std::fabs(v));
            }
        for (int k = 0; k < 3; ++k) ctx->box_lo[k] = lo[k], ctx->box_hi[k] = hi[k];
        ctx->grid_diagonal = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
        // (the view rotates about x = x0 of each rotation: a point's distance from that axis, hence its depth, stays
        // within the largest |coordinate| + |x0|; the constant below has room for both)
        ctx->coord_max = top + 2.0;
    }
    ctx->alpha_top = 0.0;
    ctx->alpha_floor = INFINITY;
    ctx->split_auto_k = 1;
    ctx->ray_depth_known = false;
    ctx->fit_known = false;
    double edge2 = 0.0;
    for (int64_t c = 0; c < n_cells; ++c) {
        if (alpha[c] > ctx->alpha_top) ctx->alpha_top = alpha[c];  // (+inf counts: it is clamped to the limit; NaN never compares greater)
        if (alpha[c] >= DBL_EPSILON && alpha[c] < ctx->alpha_floor) ctx->alpha_floor = alpha[c];
        if (alpha[c] != alpha[c]) ctx->alpha_floor = 0.0;  // (a NaN alpha: no claim about conditioning)
        const int32_t* v = cell_vert + 4 * c;
        for (int a = 0; a < 4; ++a)
            for (int b = a + 1; b < 4; ++b) {
                const double* pa = xyz + 3 * static_cast<int64_t>(v[a]);
                const double* pb = xyz + 3 * static_cast<int64_t>(v[b]);
                const double d2 = (pa[0] - pb[0]) * (pa[0] - pb[0]) + (pa[1] - pb[1]) * (pa[1] - pb[1]) + (pa[2] - pb[2]) * (pa[2] - pb[2]);
                if (d2 > edge2) edge2 = d2;
            }
    }
    ctx->edge_max = std::sqrt(edge2) * (1.0 + 1e-9);  // (the rotations round: a hair of margin)
    {   // a sphere about every 256 consecutive cells (one workgroup of build_records): kernels.hpp: GridView::block_sphere
        const int64_t n_blocks = (n_cells + 255) / 256;
        std::vector<double> sph(static_cast<size_t>(4 * n_blocks), 0.0);
        for (int64_t b = 0; b < n_blocks; ++b) {
            double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int64_t c = 256 * b; c < std::min<int64_t>(n_cells, 256 * (b + 1)); ++c)
                for (int a = 0; a < 4; ++a) {
                    const double* p = xyz + 3 * static_cast<int64_t>(cell_vert[4 * c + a]);
                    for (int k = 0; k < 3; ++k) lo[k] = std::fmin(lo[k], p[k]), hi[k] = std::fmax(hi[k], p[k]);
                }
            double r2 = 0.0;
            for (int k = 0; k < 3; ++k) {
                sph[static_cast<size_t>(4 * b + k)] = 0.5 * (lo[k] + hi[k]);
                r2 += 0.25 * (hi[k] - lo[k]) * (hi[k] - lo[k]);
            }
            sph[static_cast<size_t>(4 * b + 3)] = std::sqrt(r2) * (1.0 + 1e-12);
        }
        C5_HIP(ctx, ctx->block_sphere.ensure(sph.size() * sizeof(double) + 32));
        if (!sph.empty()) C5_HIP(ctx, hipMemcpy(ctx->block_sphere.ptr, sph.data(), sph.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    ctx->n_pts = n_pts;
    ctx->n_cells = n_cells;
    ctx->n_bfaces = static_cast<int64_t>(bfaces.size());
    ctx->grid_conforming = conforming;
    ctx->overlap_seen = false;
    return C5_OK;
}

int c5_update_scalars(c5_context* ctx, const double* alpha, const double* q, int64_t n_cells) {
    if (!ctx) return fail(nullptr, C5_ERR_INVALID, "null context");
    ++ctx->setup_epoch;  // whatever was built per view is stale ("view_cache")
    if (n_cells != ctx->n_cells) return fail(ctx, C5_ERR_INVALID, "scalar count differs from the uploaded grid");
    if (n_cells > 0 && (!alpha || !q)) return fail(ctx, C5_ERR_INVALID, "null scalar array");
    int rc = bind_device(ctx);
    if (rc) return rc;
    C5_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n_cells > 0) {
        std::vector<double> a_dev, q_dev;
        const double *a_src = alpha, *q_src = q;
        if (!ctx->cell_perm.empty()) {  // (the device keeps the cells in its own order: c5_upload_grid, "cell_order")
            a_dev.resize(static_cast<size_t>(n_cells));
            q_dev.resize(static_cast<size_t>(n_cells));
            for (size_t c = 0; c < static_cast<size_t>(n_cells); ++c) a_dev[c] = alpha[ctx->cell_perm[c]], q_dev[c] = q[ctx->cell_perm[c]];
            a_src = a_dev.data(), q_src = q_dev.data();
        }
        C5_HIP(ctx, hipMemcpy(ctx->alpha.ptr, a_src, static_cast<size_t>(n_cells) * 8, hipMemcpyHostToDevice));
        C5_HIP(ctx, hipMemcpy(ctx->q.ptr, q_src, static_cast<size_t>(n_cells) * 8, hipMemcpyHostToDevice));
    }
    ctx->alpha_top = 0.0;
    ctx->alpha_floor = INFINITY;
    for (int64_t i = 0; i < n_cells; ++i) {
        if (alpha[i] > ctx->alpha_top) ctx->alpha_top = alpha[i];
        if (alpha[i] >= DBL_EPSILON && alpha[i] < ctx->alpha_floor) ctx->alpha_floor = alpha[i];
        if (alpha[i] != alpha[i]) ctx->alpha_floor = 0.0;
    }
    return C5_OK;
}

int c5_set_solid(c5_context* ctx, int slot, const double* tets, int64_t n_tets, double colour) {
    if (!ctx) return fail(nullptr, C5_ERR_INVALID, "null context");
    if (slot < 0 || slot >= C5_MAX_SOLIDS) return fail(ctx, C5_ERR_INVALID, "solid slot %d out of range", slot);
    if (n_tets < 0 || (n_tets > 0 && !tets)) return fail(ctx, C5_ERR_INVALID, "bad solid array");
    int rc = bind_device(ctx);
    if (rc) return rc;
    C5_HIP(ctx, hipStreamSynchronize(ctx->stream));
    Solid& s = ctx->solids[slot];
    int64_t others = 0;
    for (int k = 0; k < C5_MAX_SOLIDS; ++k)
        if (k != slot) others += ctx->solids[k].n_tets;
    if (others + n_tets >= static_cast<int64_t>(c5::kNoCell))
        return fail(ctx, C5_ERR_INVALID, "solid cell count does not fit 28 bits");
    s.n_tets = n_tets;
    s.colour = colour;
    s.n_points = s.n_faces = 0;
    s.n_interior = 0;
    s.groups.clear();
    ++s.generation;  // whatever mask of its own the slot had is stale
    s.own_mask_ready = false;
    if (n_tets > 0) {
        std::vector<double> pts;
        std::vector<int32_t> faces;
        c5::unique_solid_faces(tets, n_tets, pts, faces);
        s.n_points = static_cast<int64_t>(pts.size() / 3);
        s.n_faces = static_cast<int64_t>(faces.size() / 4);
        s.n_interior = 0;
        while (s.n_interior < s.n_faces && faces[4 * static_cast<size_t>(s.n_interior) + 3] != 0) ++s.n_interior;
        {   // bounding sphere about the centre of the bounding box (no rotation makes the solid reach further)
            double lo[3] = {pts[0], pts[1], pts[2]}, hi[3] = {pts[0], pts[1], pts[2]};
            for (size_t i = 0; i < pts.size(); i += 3)
                for (int d = 0; d < 3; ++d) lo[d] = std::min(lo[d], pts[i + d]), hi[d] = std::max(hi[d], pts[i + d]);
            for (int d = 0; d < 3; ++d) s.centre[d] = 0.5 * (lo[d] + hi[d]);
            double r2 = 0.0;
            for (size_t i = 0; i < pts.size(); i += 3) {
                double q = 0.0;
                for (int d = 0; d < 3; ++d) q += (pts[i + d] - s.centre[d]) * (pts[i + d] - s.centre[d]);
                r2 = std::max(r2, q);
            }
            s.radius = std::sqrt(r2);
        }
        {   // groups of faces of about the same size (see Solid::FaceGroup), the generator's order kept inside a group:
            // consecutive cells of init_polar are angular neighbours, their faces cover neighbouring pixels
            const size_t nf = faces.size() / 4;
            std::vector<double> edge(nf);
            std::vector<int> cls(nf);
            for (size_t f = 0; f < nf; ++f) {
                const double* a = &pts[3 * static_cast<size_t>(faces[4 * f])];
                const double* b = &pts[3 * static_cast<size_t>(faces[4 * f + 1])];
                const double* c = &pts[3 * static_cast<size_t>(faces[4 * f + 2])];
                auto d2 = [](const double* u, const double* v) {
                    return (u[0] - v[0]) * (u[0] - v[0]) + (u[1] - v[1]) * (u[1] - v[1]) + (u[2] - v[2]) * (u[2] - v[2]);
                };
                edge[f] = std::sqrt(std::max(d2(a, b), std::max(d2(b, c), d2(a, c))));
                int e = -1000;
                if (edge[f] > 0.0 && std::isfinite(edge[f])) (void)std::frexp(edge[f], &e);
                cls[f] = e;
            }
            std::vector<uint32_t> order(nf);
            for (size_t f = 0; f < nf; ++f) order[f] = static_cast<uint32_t>(f);
            std::stable_sort(order.begin(), order.end(), [&](uint32_t l, uint32_t r) {
                const bool li = static_cast<int64_t>(l) < s.n_interior, ri = static_cast<int64_t>(r) < s.n_interior;
                if (li != ri) return li;
                return cls[l] > cls[r];
            });
            std::vector<int32_t> sorted(faces.size());
            s.groups.clear();
            for (size_t k = 0; k < nf; ++k) {
                const uint32_t f = order[k];
                for (int d = 0; d < 4; ++d) sorted[4 * k + d] = faces[4 * static_cast<size_t>(f) + d];
                const bool interior = static_cast<int64_t>(f) < s.n_interior;
                const bool was_interior = k > 0 && static_cast<int64_t>(order[k - 1]) < s.n_interior;
                if (k == 0 || cls[f] != cls[order[k - 1]] || interior != was_interior)
                    s.groups.push_back(Solid::FaceGroup{static_cast<int64_t>(k), 0, 0.0});
                s.groups.back().count += 1;
                s.groups.back().longest = std::max(s.groups.back().longest, edge[f]);
            }
            faces.swap(sorted);
        }
        const size_t bytes = pts.size() * sizeof(double);
        C5_HIP(ctx, s.raw.ensure(bytes));
        C5_HIP(ctx, s.faces.ensure(faces.size() * sizeof(int32_t)));
        for (int k = 0; k < (ctx->pipeline ? kFrameSlots : 1); ++k) C5_HIP(ctx, s.view[k].ensure(bytes));
        C5_HIP(ctx, hipMemcpy(s.raw.ptr, pts.data(), bytes, hipMemcpyHostToDevice));
        C5_HIP(ctx, hipMemcpy(s.faces.ptr, faces.data(), faces.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    return C5_OK;
}

int c5_set_image(c5_context* ctx, int res_x, int res_y, const double* bounds4) {
    if (!ctx) return fail(nullptr, C5_ERR_INVALID, "null context");
    ++ctx->setup_epoch;  // whatever was built per view is stale ("view_cache")
    if (!bounds4) return fail(ctx, C5_ERR_INVALID, "plane initializer. wrong manual boundaries");  // plane.cpp:262-264
    if (res_x < 2 || res_y < 2) return fail(ctx, C5_ERR_INVALID, "critical error. empty plane");
    int rc = bind_device(ctx);
    if (rc) return rc;
    rc = drain(ctx);
    if (rc) return rc;
    std::memcpy(ctx->bounds, bounds4, sizeof ctx->bounds);
    c5::ImageParams& im = ctx->im;
    im.res_x = res_x;
    im.res_y = res_y;
    im.x_min = bounds4[1];
    im.y_min = bounds4[3];
    // plane.cpp:295-302
    im.step_x = (bounds4[0] - bounds4[1]) / (static_cast<double>(res_x) - 1.);
    im.step_y = (bounds4[2] - bounds4[3]) / (static_cast<double>(res_y) - 1.);
    for (im.fit_shift = 3;; ++im.fit_shift) {  // the depth sample (DepthSamples): the finest raster of at most kFitSlots boxes
        im.fit_cols = (res_x >> im.fit_shift) + 1;
        if (static_cast<int64_t>(im.fit_cols) * ((res_y >> im.fit_shift) + 1) <= c5::kFitSlots) break;
    }
    // plane.cpp:304-314: coordinates are running sums
    std::vector<double> X(static_cast<size_t>(res_x)), Y(static_cast<size_t>(res_y));
    double cx = bounds4[1];
    for (int i = 0; i < res_x; ++i) {
        X[static_cast<size_t>(i)] = cx;
        cx = cx + im.step_x;
    }
    double cy = bounds4[3];
    for (int j = 0; j < res_y; ++j) {
        Y[static_cast<size_t>(j)] = cy;
        cy = cy + im.step_y;
    }
    C5_HIP(ctx, ctx->xtab.ensure(X.size() * 8));
    C5_HIP(ctx, ctx->ytab.ensure(Y.size() * 8));
    C5_HIP(ctx, hipMemcpy(ctx->xtab.ptr, X.data(), X.size() * 8, hipMemcpyHostToDevice));
    C5_HIP(ctx, hipMemcpy(ctx->ytab.ptr, Y.data(), Y.size() * 8, hipMemcpyHostToDevice));
    ctx->host_ytab = Y;
    ctx->have_image = true;
    rc = recompute_rows(ctx);
    if (rc) return rc;
    return ensure_image_buffers(ctx);
}

int c5_set_row_tiles(c5_context* ctx, int tile_rows, int rank, int world) {
    if (!ctx) return fail(nullptr, C5_ERR_INVALID, "null context");
    ++ctx->setup_epoch;  // whatever was built per view is stale ("view_cache")
    if (world < 1 || rank < 0 || rank >= world) return fail(ctx, C5_ERR_INVALID, "bad rank/world %d/%d", rank, world);
    if (tile_rows < 0) return fail(ctx, C5_ERR_INVALID, "bad tile_rows");
    if (world > 1 && tile_rows == 0) return fail(ctx, C5_ERR_INVALID, "tile_rows must be > 0 when world > 1");
    ctx->cfg_tile_rows = tile_rows;
    ctx->cfg_rank = rank;
    ctx->cfg_world = world;
    if (ctx->have_image) {
        int rc = bind_device(ctx);
        if (rc) return rc;
        C5_HIP(ctx, hipStreamSynchronize(ctx->stream));
        rc = recompute_rows(ctx);
        if (rc) return rc;
        return ensure_image_buffers(ctx);
    }
    return C5_OK;
}

int c5_set_row_range(c5_context* ctx, int row_begin, int row_count) {
    if (!ctx) return fail(nullptr, C5_ERR_INVALID, "null context");
    ++ctx->setup_epoch;  // whatever was built per view is stale ("view_cache")
    if (row_begin < 0 || row_count < -1) return fail(ctx, C5_ERR_INVALID, "bad row range");
    const int old_begin = ctx->cfg_row_begin, old_count = ctx->cfg_row_count;
    ctx->cfg_row_begin = row_begin;
    ctx->cfg_row_count = row_count;
    if (ctx->have_image) {
        int rc = bind_device(ctx);
        if (rc) return rc;
        C5_HIP(ctx, hipStreamSynchronize(ctx->stream));
        rc = recompute_rows(ctx);
        if (rc) {
            ctx->cfg_row_begin = old_begin;
            ctx->cfg_row_count = old_count;
            recompute_rows(ctx);
            return rc;
        }
        return ensure_image_buffers(ctx);
    }
    return C5_OK;
}

int c5_get_row_costs(c5_context* ctx, uint32_t* costs, int n_rows) {
    if (!ctx || !costs) return fail(ctx, C5_ERR_INVALID, "null argument");
    // (the option may have been switched off again since: the costs of the last frame that counted them stay readable
    // until the rows are laid out anew — a sweep probes one frame in many)
    if (!ctx->row_costs_collected) return fail(ctx, C5_ERR_STATE, "enable option \"row_costs\" before rendering");
    if (n_rows != ctx->im.n_local_rows) return fail(ctx, C5_ERR_INVALID, "expected %d rows", ctx->im.n_local_rows);
    int rc = c5_synchronize(ctx);
    if (rc) return rc;
    if (n_rows > 0)
        C5_HIP(ctx, hipMemcpy(costs, ctx->slots[ctx->row_cost_slot].row_cost.ptr, static_cast<size_t>(n_rows) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return C5_OK;
}

int c5_local_rows(const c5_context* ctx, int* n_rows) {
    if (!ctx || !n_rows) return C5_ERR_INVALID;
    *n_rows = ctx->have_image ? ctx->im.n_local_rows : 0;
    return C5_OK;
}

int c5_set_view(c5_context* ctx, const c5_rotation* rots, int n_rots) {
    if (!ctx) return fail(nullptr, C5_ERR_INVALID, "null context");
    return to_rotation_list(ctx, rots, n_rots, ctx->view);
}

int c5_set_solid_view(c5_context* ctx, int slot, const c5_rotation* rots, int n_rots) {
    if (!ctx) return fail(nullptr, C5_ERR_INVALID, "null context");
    if (slot < 0 || slot >= C5_MAX_SOLIDS) return fail(ctx, C5_ERR_INVALID, "solid slot %d out of range", slot);
    return to_rotation_list(ctx, rots, n_rots, ctx->solids[slot].rots);
}

int c5_set_alpha_limit(c5_context* ctx, double alpha_limit) {
    if (!ctx) return fail(nullptr, C5_ERR_INVALID, "null context");
    ctx->alpha_limit = alpha_limit;
    return C5_OK;
}

int c5_set_option(c5_context* ctx, const char* name, double value) {
    if (!ctx || !name) return fail(ctx, C5_ERR_INVALID, "null option");
    const std::string n(name);
    // whatever was built per view is stale ("view_cache") - except after the switches callers flip from frame to frame,
    // which the per-view data do not depend on
    if (n != "row_costs" && n != "stage_timing" && n != "walk_timing") ++ctx->setup_epoch;
    if (n == "tile") {
        if (value < 0 || value > 3) return fail(ctx, C5_ERR_INVALID, "tile must be 0, 1, 2 or 3");
        ctx->tile_shape = static_cast<int>(value);
    } else if (n == "transmittance_cutoff") {
        ctx->t_cutoff = value;
    } else if (n == "pipeline") {
        int rc = bind_device(ctx);
        if (rc) return rc;
        rc = drain(ctx);
        if (rc) return rc;
        if (ctx->n_cells > 0 || ctx->have_image)
            return fail(ctx, C5_ERR_STATE, "set \"pipeline\" before uploading the grid and setting the image");
        ctx->pipeline = static_cast<int>(value) != 0;
    } else if (n == "fuse_setup") {
        ctx->fuse_setup = static_cast<int>(value) != 0;
    } else if (n == "overlap_setup") {
        ctx->overlap_setup = static_cast<int>(value) != 0;
    } else if (n == "cell_order") {
        ctx->cell_order = static_cast<int>(value) != 0;
    } else if (n == "block_cull") {
        ctx->block_cull = static_cast<int>(value) != 0;
    } else if (n == "tile_flags") {
        ctx->tile_flags = static_cast<int>(value) != 0;
    } else if (n == "cost_order") {
        ctx->cost_order = static_cast<int>(value);  // (2: whatever the frame's size - experiments)
    } else if (n == "entry_key") {
        ctx->entry_key = static_cast<int>(value) != 0;
        ctx->overlap_seen = false;  // (with the testing value 0 an abutting entry can look like a skipped one: judge anew)
    } else if (n == "stage_slots") {
        if (value != 0 && value != 14 && value != 21) return fail(ctx, C5_ERR_INVALID, "stage_slots must be 0 (per frame), 14 or 21");
        ctx->stage_slots = static_cast<int>(value);
    } else if (n == "solid_interior_faces") {
        ctx->solid_interior_faces = static_cast<int>(value) != 0;
        for (Solid& so : ctx->solids) so.own_mask_ready = false, so.unchanged_frames = 0, so.seen_generation = ~uint64_t{0};
    } else if (n == "view_cache") {
        ctx->view_cache = static_cast<int>(value) != 0;
    } else if (n == "split_tilt_x" || n == "split_tilt_y") {  // testing: tilt of a forced split's planes
        if (!(std::fabs(value) < 64.0)) return fail(ctx, C5_ERR_INVALID, "split tilt out of range");
        (n == "split_tilt_x" ? ctx->split_tilt_x : ctx->split_tilt_y) = value;
    } else if (n == "entry_records") {
        ctx->entry_records = static_cast<int>(value) != 0;
    } else if (n == "depth_split") {
        if (value < 0 || value > c5::kMaxSlabs || value != std::floor(value))
            return fail(ctx, C5_ERR_INVALID, "depth_split must be 0 (per frame), 1 (never) or 2..%d slabs", c5::kMaxSlabs);
        ctx->depth_split = static_cast<int>(value);
    } else if (n == "solid_cache") {
        ctx->solid_cache = static_cast<int>(value) != 0;
        for (Solid& so : ctx->solids) so.own_mask_ready = false, so.unchanged_frames = 0, so.seen_generation = ~uint64_t{0};
    } else if (n == "algorithm") {
        if (value != 0 && value != 1) return fail(ctx, C5_ERR_INVALID, "algorithm must be 0 (walk) or 1 (bin_sort_resolve)");
        ctx->algorithm = static_cast<int>(value);
    } else if (n == "lds_stage") {
        ctx->lds_stage = static_cast<int>(value) < 0 ? 0 : (static_cast<int>(value) > 2 ? 2 : static_cast<int>(value));
    } else if (n == "integration") {
        ctx->order = static_cast<int>(value) != 0;
    } else if (n == "entry_pool") {  // testing: (re)size the overflow pool of the entry lists, in records
        if (value < 1 || value > 16777214) return fail(ctx, C5_ERR_INVALID, "entry_pool out of range");
        int rc = drain(ctx);
        if (rc) return rc;
        for (FrameSlot& fs : ctx->slots) {
            fs.entry_capacity = static_cast<int64_t>(value);
            C5_HIP(ctx, fs.pool.ensure(static_cast<size_t>(fs.entry_capacity) * sizeof(c5::Entry)));
        }
    } else if (n == "band_rows") {  // tuning: image rows per XCD band of the walk (0: default 32)
        if (value < 0 || value > 4096) return fail(ctx, C5_ERR_INVALID, "band_rows out of range");
        ctx->band_rows = static_cast<int>(value);
    } else if (n == "lds_pad") {  // tuning: occupancy experiments (scripts/occupancy_sweep.py)
        if (value < 0 || value > 96 * 1024) return fail(ctx, C5_ERR_INVALID, "lds_pad out of range");
        ctx->lds_pad = static_cast<int>(value);
    } else if (n == "xcd_mode") {
        ctx->xcd_mode = static_cast<int>(value) < 0 ? 0 : (static_cast<int>(value) > 2 ? 2 : static_cast<int>(value));
    } else if (n == "row_costs") {
        ctx->row_costs = static_cast<int>(value) != 0;
    } else if (n == "stage_timing") {
        ctx->stage_timing = static_cast<int>(value) != 0;
    } else if (n == "walk_timing") {
        if (!(value >= 0.0 && value <= 1024.0)) return fail(ctx, C5_ERR_INVALID, "walk_timing: 0 (off) or every N-th launch, N <= 1024");
        ctx->walk_timing = static_cast<int>(value);
        ctx->walk_seq = 0;
    } else {
        return fail(ctx, C5_ERR_INVALID, "unknown option '%s'", name);
    }
    return C5_OK;
}

int c5_render_device(c5_context* ctx, void* out_device) {
    if (!ctx) return fail(nullptr, C5_ERR_INVALID, "null context");
    if (!out_device) return fail(ctx, C5_ERR_INVALID, "null output pointer");
    return enqueue_frame(ctx, static_cast<float2*>(out_device));
}

int c5_synchronize(c5_context* ctx) {
    if (!ctx) return fail(nullptr, C5_ERR_INVALID, "null context");
    int rc = bind_device(ctx);
    if (rc) return rc;
    return wait_and_collect(ctx);
}

namespace {
bool is_pinned(const void* p) {
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();  // an ordinary (pageable) pointer: not an error here
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

// Device image -> host.  Pinned destination: one asynchronous copy.  Pageable destination: 4 MB chunks
// through two pinned staging buffers, each copied out on the host threads while the next is in flight
// (a plain hipMemcpy into pageable memory ran at 7.7 GB/s: 4.5 ms for a 2400x1800 image).
int copy_image_to_host(c5_context* ctx, const void* dev, void* host, size_t bytes) {
    hipStream_t cs = ctx->copy_stream;
    if (is_pinned(host)) {
        C5_HIP(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, cs));
        C5_HIP(ctx, hipStreamSynchronize(cs));
        return C5_OK;
    }
    for (int k = 0; k < 2; ++k)
        if (!ctx->stage[k]) C5_HIP(ctx, hipHostMalloc(&ctx->stage[k], kStageChunk, hipHostMallocDefault));
    const size_t n_chunks = (bytes + kStageChunk - 1) / kStageChunk;
    for (size_t i = 0; i <= n_chunks; ++i) {
        if (i < n_chunks) {
            const size_t off = i * kStageChunk, n = std::min(kStageChunk, bytes - off);
            C5_HIP(ctx, hipMemcpyAsync(ctx->stage[i & 1], static_cast<const char*>(dev) + off, n, hipMemcpyDeviceToHost, cs));
            C5_HIP(ctx, hipEventRecord(ctx->stage_ev[i & 1], cs));
        }
        if (i > 0) {
            const size_t off = (i - 1) * kStageChunk, n = std::min(kStageChunk, bytes - off);
            C5_HIP(ctx, hipEventSynchronize(ctx->stage_ev[(i - 1) & 1]));
            c5::parallel_copy(static_cast<char*>(host) + off, ctx->stage[(i - 1) & 1], n);
        }
    }
    return C5_OK;
}
}  // namespace

int c5_render(c5_context* ctx, float* out_host) {
    if (!ctx) return fail(nullptr, C5_ERR_INVALID, "null context");
    if (!out_host) return fail(ctx, C5_ERR_INVALID, "null output pointer");
    if (ctx->hr_count) return fail(ctx, C5_ERR_STATE, "c5_render while c5_render_host_async frames are outstanding");
    for (int attempt = 0; attempt < 3; ++attempt) {
        int rc = enqueue_frame(ctx, ctx->out.as<float2>());
        if (rc) return rc;
        rc = c5_synchronize(ctx);
        if (rc == C5_RETRY) continue;
        if (rc) return rc;
        const size_t bytes = static_cast<size_t>(ctx->im.n_local_rows) * ctx->im.res_x * 2 * sizeof(float);
        return copy_image_to_host(ctx, ctx->out.ptr, out_host, bytes);
    }
    return fail(ctx, C5_ERR_STATE, "entry buffer kept overflowing");
}

namespace {
// Copy the local strip (device) to the host: either as it is, or every row tile to its place in the full image.
int enqueue_strip_copy(c5_context* ctx, const void* strip, float* host, bool into_full_frame) {
    const c5::ImageParams& im = ctx->im;
    const size_t row_bytes = static_cast<size_t>(im.res_x) * 2 * sizeof(float);
    hipStream_t cs = ctx->copy_stream;
    if (!into_full_frame) {
        C5_HIP(ctx, hipMemcpyAsync(host, strip, row_bytes * im.n_local_rows, hipMemcpyDeviceToHost, cs));
        return C5_OK;
    }
    char* const frame = reinterpret_cast<char*>(host);
    const char* const src = static_cast<const char*>(strip);
    if (im.world == 1) {  // one contiguous block of rows
        C5_HIP(ctx, hipMemcpyAsync(frame + row_bytes * im.row_begin, src, row_bytes * im.n_local_rows, hipMemcpyDeviceToHost, cs));
        return C5_OK;
    }
    // cyclic tiles: local tile t is global tile t * world + rank (counted from row_begin): ONE 2-D copy for the
    // whole tiles (a "row" of the 2-D copy = one tile of tile_rows image rows) and one for a short last tile
    const size_t tile_bytes = row_bytes * im.tile_rows;
    const int whole = im.n_local_rows / im.tile_rows, rest = im.n_local_rows - whole * im.tile_rows;
    char* const first = frame + row_bytes * im.row_begin + tile_bytes * im.rank;
    if (whole > 0)
        C5_HIP(ctx, hipMemcpy2DAsync(first, tile_bytes * im.world, src, tile_bytes, tile_bytes, static_cast<size_t>(whole),
                                     hipMemcpyDeviceToHost, cs));
    if (rest > 0)
        C5_HIP(ctx, hipMemcpyAsync(first + tile_bytes * im.world * whole, src + tile_bytes * whole, row_bytes * rest,
                                   hipMemcpyDeviceToHost, cs));
    return C5_OK;
}

int render_host_async(c5_context* ctx, float* out_host, bool into_full_frame) {
    if (!ctx) return fail(nullptr, C5_ERR_INVALID, "null context");
    if (!out_host) return fail(ctx, C5_ERR_INVALID, "null output pointer");
    if (ctx->hr_count >= C5_HOST_RING)
        return fail(ctx, C5_ERR_STATE, "%d frames outstanding: call c5_render_host_wait first", C5_HOST_RING);
    int rc = bind_device(ctx);
    if (rc) return rc;
    c5_context::HostFrame& h = ctx->hring[ctx->hr_next];
    const size_t bytes = static_cast<size_t>(ctx->im.n_local_rows) * ctx->im.res_x * 2 * sizeof(float);
    C5_HIP(ctx, h.img.ensure(((bytes + 8191) / 8192) * 8192));
    C5_HIP(ctx, h.counters.ensure(kCountersBytes));
    // (the slot's previous copy is complete: its c5_render_host_wait has returned)
    rc = enqueue_frame(ctx, h.img.as<float2>(), h.counters.as<c5::FrameCounters>());
    if (rc) return rc;
    C5_HIP(ctx, hipEventRecord(h.rendered, ctx->stream));
    C5_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, h.rendered, 0));
    rc = enqueue_strip_copy(ctx, h.img.ptr, out_host, into_full_frame);
    if (rc) return rc;
    // THIS frame's failure words (walk_overflow, entry_overflow: adjacent in shard 0 of its own counters, cleared by its
    // own first kernel): the wait can tell without touching the render stream, and no frame enqueued later adds to them.
    // (Rounds 2-3 copied the context's cumulative sticky words here: a snapshot that depended on how a NULL-stream
    // hipMemset of c5_create was ordered against this copy stream, and that a later frame's raster could add to.)
    static_assert(offsetof(c5::FrameCounters, entry_overflow) == offsetof(c5::FrameCounters, walk_overflow) + sizeof(unsigned) &&
                  offsetof(c5::FrameCounters, overlap_rays) == offsetof(c5::FrameCounters, walk_overflow) + 2 * sizeof(unsigned), "status = adjacent words");
    C5_HIP(ctx, hipMemcpyAsync(h.status, reinterpret_cast<const char*>(h.counters.ptr) + offsetof(c5::FrameCounters, walk_overflow),
                               kStatusWords * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->copy_stream));
    C5_HIP(ctx, hipEventRecord(h.copied, ctx->copy_stream));
    ctx->hr_next = (ctx->hr_next + 1) % C5_HOST_RING;
    ctx->hr_count += 1;
    return C5_OK;
}
}  // namespace

int c5_render_host_async(c5_context* ctx, float* out_host) { return render_host_async(ctx, out_host, false); }
int c5_render_frame_rows_async(c5_context* ctx, float* frame_host) { return render_host_async(ctx, frame_host, true); }

int c5_render_host_wait(c5_context* ctx) {
    if (!ctx) return fail(nullptr, C5_ERR_INVALID, "null context");
    if (ctx->hr_count == 0) return fail(ctx, C5_ERR_STATE, "no c5_render_host_async frame is outstanding");
    int rc = bind_device(ctx);
    if (rc) return rc;
    c5_context::HostFrame& h = ctx->hring[ctx->hr_head];
    C5_HIP(ctx, hipEventSynchronize(h.copied));
    ctx->hr_head = (ctx->hr_head + 1) % C5_HOST_RING;
    ctx->hr_count -= 1;
    if (ctx->hr_retry_left > 0) {  // enqueued before an overflow was noticed: rendered with the buffers that were too small
        ctx->hr_retry_left -= 1;
        return fail(ctx, C5_RETRY, "frame was enqueued before an internal buffer was grown: render it again");
    }
    if (h.status[0] == 0 && h.status[1] == 0 && h.status[2] == 0) return C5_OK;
    // this frame failed: settle it (waits for the render stream, grows what was too small); every frame enqueued behind
    // it used the same buffers
    const unsigned lost = h.status[0], refused = h.status[1], overlapping = h.status[2];
    ctx->hr_retry_left = ctx->hr_count;
    rc = wait_and_collect(ctx);
    if (rc == C5_OK)
        rc = fail(ctx, C5_RETRY, "frame incomplete (%u boundary entries without a pool slot, %u rays over the step bound, %u rays through "
                  "interpenetrating cells by its own counters; the context's cumulative words read %u / %u / %u when the stream was waited "
                  "for; pool now %lld records): render again",
                  refused, lost, overlapping, ctx->host_sticky[0], ctx->host_sticky[1], ctx->host_sticky[2],
                  static_cast<long long>(ctx->slots[0].entry_capacity));
    return rc;
}

int c5_host_alloc(c5_context* ctx, size_t bytes, void** out_ptr) {
    if (!ctx || !out_ptr) return fail(ctx, C5_ERR_INVALID, "null argument");
    int rc = bind_device(ctx);
    if (rc) return rc;
    *out_ptr = nullptr;
    // portable: several contexts (one per GPU) copy their rows into the same frame (c5_render_frame_rows_async)
    C5_HIP(ctx, hipHostMalloc(out_ptr, bytes ? bytes : 1, hipHostMallocPortable));
    return C5_OK;
}

int c5_host_free(c5_context* ctx, void* ptr) {
    if (!ctx) return fail(nullptr, C5_ERR_INVALID, "null context");
    if (!ptr) return C5_OK;
    C5_HIP(ctx, hipHostFree(ptr));
    return C5_OK;
}

int c5_get_stats(c5_context* ctx, c5_stats* out) {
    if (!ctx || !out) return fail(ctx, C5_ERR_INVALID, "null argument");
    int rc = c5_synchronize(ctx);
    *out = ctx->last;
    return rc;
}

int c5_walk_kernel_ms(c5_context* ctx, int reset, double* avg_ms, int64_t* launches) {
    if (!ctx) return fail(nullptr, C5_ERR_INVALID, "null context");
    int rc = bind_device(ctx);
    if (rc) return rc;
    C5_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < ctx->walk_used; ++k) {
        float ms = 0.f;
        C5_HIP(ctx, hipEventElapsedTime(&ms, ctx->walk_a[k], ctx->walk_b[k]));
        ctx->walk_ms_sum += ms;
    }
    ctx->walk_launches += ctx->walk_used;
    ctx->walk_used = 0;
    if (avg_ms) *avg_ms = ctx->walk_launches ? ctx->walk_ms_sum / static_cast<double>(ctx->walk_launches) : 0.0;
    if (launches) *launches = ctx->walk_launches;
    if (reset) {
        ctx->walk_ms_sum = 0.0;
        ctx->walk_launches = 0;
    }
    return C5_OK;
}

int c5_face_adjacency(const int32_t* cell_vert, int64_t n_cells, int64_t n_pts, int32_t* adj,
                      int64_t* n_boundary_faces) {
    if (n_cells < 0 || n_pts < 0 || (n_cells > 0 && (!cell_vert || !adj)))
        return fail(nullptr, C5_ERR_INVALID, "bad adjacency arguments");
    std::vector<int32_t> a;
    std::vector<uint32_t> b;
    std::string err;
    if (!c5::build_face_adjacency(cell_vert, n_cells, n_pts, a, b, err))
        return fail(nullptr, err.find("range") != std::string::npos ? C5_ERR_INVALID : C5_ERR_MESH, "%s", err.c_str());
    if (n_cells > 0) std::memcpy(adj, a.data(), a.size() * sizeof(int32_t));
    if (n_boundary_faces) *n_boundary_faces = static_cast<int64_t>(b.size());
    return C5_OK;
}

int c5_weld_points(const double* xyz, int64_t n_pts, int32_t* rep, int64_t* n_merged) {
    if (n_pts < 0 || (n_pts > 0 && (!xyz || !rep))) return fail(nullptr, C5_ERR_INVALID, "bad weld arguments");
    for (int64_t i = 0; i < 3 * n_pts; ++i)
        if (!std::isfinite(xyz[i])) return fail(nullptr, C5_ERR_INVALID, "point %lld has a non-finite coordinate", static_cast<long long>(i / 3));
    std::vector<int32_t> r;
    const int64_t m = c5::weld_points(xyz, n_pts, r);
    if (n_pts > 0) std::memcpy(rep, r.data(), r.size() * sizeof(int32_t));
    if (n_merged) *n_merged = m;
    return C5_OK;
}

int c5_download_view_points(c5_context* ctx, double* xyz) {
    if (!ctx || !xyz) return fail(ctx, C5_ERR_INVALID, "null argument");
    const int frame_rc = c5_synchronize(ctx);
    if (frame_rc && frame_rc != C5_RETRY) return frame_rc;
    const size_t n = static_cast<size_t>(ctx->n_pts);
    std::vector<double> x(n), y(n), z(n);
    if (n) {
        C5_HIP(ctx, hipMemcpy(x.data(), ctx->slots[ctx->last_slot].vx.ptr, n * 8, hipMemcpyDeviceToHost));
        C5_HIP(ctx, hipMemcpy(y.data(), ctx->slots[ctx->last_slot].vy.ptr, n * 8, hipMemcpyDeviceToHost));
        C5_HIP(ctx, hipMemcpy(z.data(), ctx->slots[ctx->last_slot].vz.ptr, n * 8, hipMemcpyDeviceToHost));
    }
    for (size_t i = 0; i < n; ++i) {
        xyz[3 * i] = x[i];
        xyz[3 * i + 1] = y[i];
        xyz[3 * i + 2] = z[i];
    }
    return frame_rc;  // the points are valid either way; C5_RETRY says the frame they belong to is not
}

}  // extern "C"
