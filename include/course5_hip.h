/*
 * course5_hip.h — C ABI of the MI355X (gfx950) render path for mlozhechko/course5.
 *
 * The reference has no FFI: its seam is three C++ calls made once per frame from
 * project/src/main.cpp:127-129,
 *
 *     plane base_plane{res_x, res_y, {acc_disk, roche_lobe, acc_sphere}, domain};   // plane.cpp:260-315
 *     base_plane.find_intersections();                                              // plane.cpp:184-192
 *     object2d result = base_plane.trace_rays(tetra_value::alpha, tetra_value::Q);  // plane.cpp:144-172
 *
 * preceded by the view transform object3d_base::rotate_around_{x,y}_axis
 * (object3d_base.cpp:202-219, main.cpp:105-107,112-114).  This header is what a maintainer
 * binds in their place (INTEGRATION.md shows the edit).  Everything is extern "C" with plain
 * pointers and sizes; no C++ types, no exceptions and no torch types cross it.
 *
 * Conventions
 *   - every function returns an int status (C5_OK == 0); c5_last_error() gives the message
 *     (the reference throws std::runtime_error instead: plane.cpp:40,152,263,270, line.cpp:45);
 *   - a context is used by one host thread at a time and drives one GPU;
 *   - caller keeps ownership of every host array; the library copies what it needs;
 *   - image layout is out[row][col][2] fp32, col fastest, channel innermost — the order
 *     object2d::export_to_vti writes (object2d.cpp:17-21); channel 0 = tau
 *     (line.cpp:176-193), channel 1 = I (line.cpp:195-227).
 */
#ifndef COURSE5_HIP_H
#define COURSE5_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C5_ABI_VERSION 2

enum {
    C5_OK = 0,
    C5_ERR_INVALID = 1,     /* bad argument (null pointer, size, id range) */
    C5_ERR_STATE = 2,       /* call order: render before grid/image were set */
    C5_ERR_HIP = 3,         /* HIP runtime failure (message carries hipGetErrorString) */
    C5_ERR_MESH = 4,        /* c5_face_adjacency: a face is shared by more than two cells */
    C5_ERR_NO_DEVICE = 5,   /* no usable GPU */
    C5_ERR_WALK = 6,        /* a ray exceeded the step bound (malformed grid) */
    C5_RETRY = 7            /* EVERY frame enqueued since the last call that waited for the stream is incomplete or
                               wrong and must not be used; call c5_render_device again.  Two causes, both settled by the
                               library before it says so: an internal buffer was too small and has been grown; or the
                               walk met rays that had to SKIP a boundary entry inside a stretch of cells they had walked
                               - components of the grid that share no face interpenetrate, which the reference simply
                               bins and sorts (plane.cpp:184-192, line.cpp:138) - and the grid is rendered with
                               "algorithm" 1 from now on.  Reported by every call that waits for the stream
                               (c5_synchronize, c5_get_stats, c5_set_stream, c5_get_row_costs,
                               c5_download_view_points, c5_render_host_wait); c5_render retries by itself. */
};

#define C5_MAX_ROTATIONS 8
#define C5_MAX_SOLIDS 8

typedef struct c5_context c5_context;

/* One elementary in-place rotation, applied in list order to every vertex.
 * axis 0: tetra::point_rotate_around_x_axis (tetra.cpp:44-48);
 * axis 1: tetra::point_rotate_around_y_axis about the line x = x0, z = 0 (tetra.cpp:51-62).
 * The host evaluates cos/sin (libm, like the reference); the device applies them with
 * separately rounded multiplies and adds, so transformed vertices equal the reference's. */
typedef struct c5_rotation {
    int32_t axis;
    int32_t reserved;
    double angle; /* radians */
    double x0;
} c5_rotation;

typedef struct c5_stats {
    int64_t segments;        /* ray-tet segments with dz > 0 == plane::count_all_intersections (plane.cpp:3-12) */
    int64_t covered_pixels;  /* pixels with at least one segment */
    int64_t solid_pixels;    /* pixels overwritten by a solid colour (line.cpp:246-249) */
    int64_t entries;         /* boundary entry records produced by the entry raster */
    int64_t boundary_faces;  /* static: faces with no neighbour */
    int64_t steps;           /* walk steps taken (>= segments) */
    int32_t walk_overflow;   /* rays that hit the step bound */
    int32_t entry_overflow;  /* the overflow pool was too small for this frame (C5_RETRY) */
    /* GPU time of the last frame per stage, milliseconds (HIP events on the context stream) */
    float ms_transform;      /* view transform                     (a2); this and the next two are exactly 0 for a frame that
                              * reused the per-view data of the frames before it (option "view_cache") */
    float ms_records;        /* per-cell walk records              (a1, a10) */
    float ms_entries;        /* boundary entry raster, one pass    (a6/a7 for boundary faces) */
    float ms_solids;         /* solid mask raster                  (a6, a9) */
    float ms_walk;           /* walk_composite                     (a11-a14) */
    float ms_total;          /* first kernel start -> image complete in HBM */
    int64_t odd_pixels;      /* bin_sort_resolve only: (pixel, cell) pairs covered by an odd number of the
                                cell's faces, i.e. exactly degenerate alignment; the reference mis-pairs or
                                aborts there (plane.cpp:39-41, line.cpp:40-47), here they are skipped */
    int64_t pool_entries;    /* second and further entries of rays this frame needed room for (sum over the
                                pixels of entries - 1): a property of grid, view and image alone */
    int64_t pool_capacity;   /* room there was; the frame is complete iff pool_entries <= pool_capacity */
} c5_stats;

/* --- lifetime ----------------------------------------------------------------------------- */
int c5_abi_version(void);
int c5_device_count(int* count);
int c5_create(int device_ordinal, c5_context** out_ctx);
void c5_destroy(c5_context* ctx);
/* Message of the last failure on ctx (or of the last c5_create failure when ctx == NULL). */
const char* c5_last_error(const c5_context* ctx);

/* Run the context's work on a caller-owned HIP stream (hipStream_t passed as void*), e.g. the
 * stream of the framework that owns the output buffer, so that ordering with the caller's own
 * kernels and collectives needs no host synchronisation.  NULL restores the context's own stream. */
int c5_set_stream(c5_context* ctx, void* hip_stream);  /* waits for the old stream first: may return C5_RETRY */

/* --- scene (persistent across frames) ------------------------------------------------------ */
/* Volume grid: replaces object3d_base::read_vtk_file's per-cell copies (object3d_base.cpp:13-53)
 * and the tetra AoS (tetra.hpp:12-46).  xyz[n_pts][3] raw (untransformed) points,
 * cell_vert[n_cells][4] point ids, alpha/q[n_cells] = AbsorpCoef / radEnLooseRate
 * (object3d_accretion_disk.cpp:4).  Points with equal coordinates are welded (c5_weld_points), then the
 * face adjacency is built; n_cells must be < 2^28 (line.hpp:71-79).  A grid in which some face belongs to more than two cells cannot be walked:
 * it is accepted and rendered with "algorithm" 1 (see c5_set_option); so is, from the first frame that shows it, a grid
 * whose components share no face but interpenetrate (C5_RETRY once, see above). */
int c5_upload_grid(c5_context* ctx, const double* xyz, int64_t n_pts, const int32_t* cell_vert,
                   int64_t n_cells, const double* alpha, const double* q);
/* Replace only the cell scalars of the uploaded grid. */
int c5_update_scalars(c5_context* ctx, const double* alpha, const double* q, int64_t n_cells);
/* Solid object `slot` (0..C5_MAX_SOLIDS-1): tets[n][4][3] raw vertex copies, one colour.
 * Replaces the solid part of the tetra vector (main.cpp:110-116,127; plane.cpp:130-131).
 * n == 0 removes the object.  Higher slots / higher tet index win ties, like serial -j1. */
int c5_set_solid(c5_context* ctx, int slot, const double* tets, int64_t n_tets, double colour);

/* --- per-frame parameters -------------------------------------------------------------------- */
/* plane::plane(res_x, res_y, ..., bounds) (plane.cpp:260-315); bounds4 in the reference's order
 * {x_max, x_min, y_max, y_min} (main.cpp:83). */
int c5_set_image(c5_context* ctx, int res_x, int res_y, const double* bounds4);
/* Row sharding for multi-GPU: rows are grouped in tiles of tile_rows; tile t belongs to
 * rank t % world.  Default (1 tile of res_y rows, world 1) renders the whole image.  The local
 * strip holds this rank's rows in ascending global order. */
int c5_set_row_tiles(c5_context* ctx, int tile_rows, int rank, int world);
/* Restrict rendering to the contiguous rows [row_begin, row_begin + row_count) (row_count -1 = to the
 * end); tiles of c5_set_row_tiles are then counted from row_begin.  A context only builds the
 * per-view records of cells its rows can reach, so contiguous blocks also shard the per-view
 * setup.  Default: the whole image. */
int c5_set_row_range(c5_context* ctx, int row_begin, int row_count);
int c5_local_rows(const c5_context* ctx, int* n_rows);
/* Segments per local row of the last frame rendered with option "row_costs" = 1 (the option may be off again since):
 * the cost estimate for balancing row blocks across GPUs. */
int c5_get_row_costs(c5_context* ctx, uint32_t* costs, int n_rows);
/* View transform of the volume grid (main.cpp:105-107) and of each solid (main.cpp:112-114,
 * object3d_roche_lobe.cpp:48). */
int c5_set_view(c5_context* ctx, const c5_rotation* rots, int n_rots);
int c5_set_solid_view(c5_context* ctx, int slot, const c5_rotation* rots, int n_rots);
/* app::config.limit_alpha_value (config.hpp:25, line.cpp:204,216-218); default 2.5. */
int c5_set_alpha_limit(c5_context* ctx, double alpha_limit);
/* Knobs:
 *   "integration"  0 (default): rays are walked from -z to +z and ch1 is integrated back to front
 *                  with the reference's own recurrence and rounding (line.cpp:206-225);
 *                  1: front to back from the viewer, I = sum T_k S_k, with the wavefront early-out
 *                  once the transmittance T falls below "transmittance_cutoff" (default 1e-12,
 *                  0 disables).  Both agree to rounding wherever the reference's recurrence is
 *                  well conditioned (it is not for DBL_EPSILON <= alpha < ~1e-8).
 *   "depth_split"  0 (default): a frame whose rays do not fill the GPU's wavefront slots - a small image, one GPU's rows of
 *                  a frame - and are long enough is rendered with every ray cut at K - 1 parallel planes of depth (K <= 4,
 *                  chosen from the statistics of the frame before, the planes tilted so that an oblique view's rays are cut
 *                  at equal fractions: fitted to where a sample of that frame's rays entered the grid and where they
 *                  ended): K jobs per 8x8 pixel tile walk the K parts at once and
 *                  the partial integrals are composed in depth order (tau = sum; I <- exp(-tauc_s) I + b_s: the recurrence
 *                  of line.cpp:206-225 is affine in I).  Same segment counts; I differs from the whole-ray walk in its
 *                  rounding ORDER only (~1e-16) - which is why a grid with a clamped alpha in [DBL_EPSILON, 1e-6), where
 *                  the reference's recurrence is dominated by its own cancellation error, is never cut.  1: never.
 *                  2..8: always that many slabs, with planes of constant depth that depend on the view alone (renders of
 *                  different rows of one frame are bit-equal only at the same slab count; "split_tilt_x" / "split_tilt_y",
 *                  testing: the planes' tilt, depth - tx x - ty y = const).  Default tile shape, "lds_stage" 2,
 *                  "integration" 0 and "xcd_mode" 2 only; whole rays otherwise.  DESIGN.md section 4.3.
 *   "algorithm"    0 (default for conforming grids): face-adjacency walk.  1: bin_sort_resolve, the
 *                  reference's own algorithm on the GPU (every face of every cell scan-converted onto
 *                  the pixels, per-pixel sort by z, integrate) — handles tet soups, overlapping and
 *                  non-conforming cells like the reference does; c5_upload_grid selects it by itself
 *                  when a face is shared by more than two cells.  Needs ~24 B per ray-cell segment
 *                  and synchronises inside c5_render_device.
 *   "lds_stage"    2 (default): walk_composite_lds with LDS-DMA staging — per step a wavefront loads each distinct
 *                  cell record once, straight into LDS (global_load_lds_dwordx4), and its rays read it from there;
 *                  1: the same staged through vector registers (global_load + ds_write_b128; also what 2 falls
 *                  back to beyond 2^24 cells); 0: every lane loads its own record.  Same results, bit for bit.
 *   "cost_order"   1 (default): frames with fewer rays than about two rounds of the GPU's wavefront slots (and more than ~1 500
 *                  wavefronts) start the rows of their image that hold the LONGEST rays first (by the last frame the caller
 *                  waited for) instead of top to bottom: a launch ends on whatever started last; 0: always top to bottom;
 *                  2: that order whatever the frame's size (experiments).  Same results.
 *   "cell_order"   1 (default; read by c5_upload_grid: set it BEFORE the upload): the library keeps the cells in Morton order of
 *                  their centroids, whatever order the caller has them in (grids of 4 096 cells and more that go to the
 *                  walk); c5_update_scalars still takes its arrays in the caller's order.  Same results, bit for bit.
 *   "block_cull"   1 (default): a context that renders a part of the image's rows judges every 256 consecutive cells by a
 *                  sphere about them before it builds their per-view records; 0: cell by cell only.  Same results.
 *   "tile_flags"   1 (default): the entry raster marks the 8x8 pixel tiles a boundary face's box meets, and a wavefront of the
 *                  walk looks at its tile's mark before anything else (default tile shape, no solids); 0: every wavefront
 *                  reads its pixels' entry heads.  Same results.
 *   "stage_slots"  "lds_stage" 1 / 2: distinct cells staged per wavefront and step (LDS-DMA passes of seven).  0 (default): 21
 *                  when the frame before had fewer than 120 ray-cell segments per cell (pixels coarse against the cells:
 *                  more distinct cells per 8x8 tile), else 14 (one more wavefront per SIMD); 14 / 21: fixed.  Same results
 *                  either way.
 *   "tile"         wavefront tile: 0 = 64x1 row tile, 1 = 16x4, 2 = 8x8 pixels, four wavefronts per workgroup; 3 (default):
 *                  8x8 pixels, one wavefront per workgroup (its slot is free again when ITS rays are done).
 *   "xcd_mode"     how workgroups map to the 8 XCDs (blocks b and b + 8 share an L2): 2 (default): square
 *                  super-blocks of workgroups dealt round-robin; 1: bands of image rows; 0: row-major tiles.
 *   "band_rows"    tuning: rows per super-block ("xcd_mode" 2, 0 = default 32) or band (1, default 16).
 *   "fuse_setup"   1: the per-cell records and the boundary entry lists are built by ONE launch of interleaved
 *                  workgroups; 0 (default): two launches.  Same results; measured slower fused.
 *   "solid_cache"  1 (default): a solid whose view and image are the same as in the frame before (the accretor sphere
 *                  never rotates, main.cpp:116; in a -D sweep only the lobe moves) is rastered once into a mask of its
 *                  own, which later frames lay over theirs; 0: every solid is transformed and rastered every frame.
 *                  Same masks either way.
 *   "solid_interior_faces"  0 (default): a face of a solid with a cell of non-zero volume on either side is not rastered: the
 *                  mask is the union over all faces (plane.cpp:130-131 -> line.cpp:246-249), and a ray through such a face
 *                  also meets a face with nothing behind it — the three fan faces of every cell of the reference's
 *                  centre-fan solids (object3d_base.cpp:152-193).  Only where get_pixel_by_x/_y's clamp (plane.cpp:194-212)
 *                  has no hand in the face's pixels: an interior face within a pixel of a border, or beyond it, is kept.
 *                  1 (testing): every unique face is rastered.  Same masks, bit for bit.
 *   "view_cache"   1 (default): a frame whose grid, scalars, image, rows, view, alpha limit and order are those of the TWO frames
 *                  before it reuses their per-view data — transformed vertices, cell records, boundary entry lists — instead
 *                  of building them again: the persistent device grid of a donor sweep (main.cpp:112-116: only the lobe
 *                  turns).  The second frame of such a run builds everything once more and tells its walk to leave the
 *                  per-pixel entry heads in place (the walk normally hands them back cleared); the third and later ones
 *                  skip the three setup launches (c5_stats: their ms_transform / ms_records / ms_entries are exactly 0).
 *                  Anything the data depend on makes them stale: c5_upload_grid, c5_update_scalars, c5_set_image, the row
 *                  setters, c5_set_stream, any option but "row_costs" / "stage_timing" / "walk_timing", a grown entry pool.
 *                  A sweep whose view changes every frame never pays for it.  0: every frame builds its own.  Same results.
 *   "overlap_setup" 1: entry lists and solid mask are built on a side stream while build_records
 *                  runs (only when "stage_timing" is 0).  Default 0: measured no faster.
 *   "pipeline"     1: two frame slots; the per-view setup of frame k + 1 runs on a second stream while
 *                  frame k is walked (set before c5_upload_grid / c5_set_image).  Default 0; measured
 *                  2 % faster on the C3 frame at the end of round 1 (a second set of per-view records).
 *   "entry_pool"   testing: size of the overflow pool of the per-pixel entry lists, in records (it holds
 *                  the second and further entries of a ray).  A frame is complete iff its total demand
 *                  (c5_stats.pool_entries) fits; the library keeps the pool at twice the demand of the
 *                  last frame it looked at and reports C5_RETRY for frames that did not fit.
 *   "lds_pad"      tuning: extra dynamic LDS per workgroup in bytes, to cap the resident wavefronts.
 *   "row_costs"    1: walk_composite also accumulates segments per image row (c5_get_row_costs).  May be switched per
 *                  frame: the costs of the last frame that counted them stay readable until the rows are laid out
 *                  anew (a sweep probes one frame in many).
 *   "entry_key"    1 (default): a boundary entry is keyed a slack behind its face, the same slack for every face of the
 *                  frame (+ more for faces steep against the rays), so that a ray leaving through a face that has no
 *                  partner with the same three points — hanging nodes: a coarse face against several fine ones — is
 *                  picked up by the abutting cell (the reference never looks at connectivity, object3d_base.cpp:37-42;
 *                  DESIGN.md section 5).  0 (testing): keyed at the face's own depth, as before round 3.
 *   "stage_timing" / "walk_timing"  record HIP events per stage (0 / 1; c5_stats::ms_* are those of the last frame rendered
 *                  with it on) / around walk_composite ("walk_timing" N: around every N-th launch, for c5_walk_kernel_ms).
 *                  Both default to 0: the six stage events cost 21-25 us of a 0.52-ms frame when frames follow one another
 *                  without a wait, the two around the walk 6. */
int c5_set_option(c5_context* ctx, const char* name, double value);

/* --- render ---------------------------------------------------------------------------------- */
/* find_intersections + trace_rays for the local rows.  out_host[local_rows][res_x][2].  Synchronous; retries
 * by itself on C5_RETRY.  A pinned out_host (c5_host_alloc) receives the image by one direct copy; a pageable
 * one through pinned staging chunks, copied out on the host threads while the next chunk is in flight. */
int c5_render(c5_context* ctx, float* out_host);
/* Same, asynchronous on the context's stream, into device memory (hipMalloc'ed by anyone in
 * this process).  Pair with c5_synchronize. */
int c5_render_device(c5_context* ctx, void* out_device);
int c5_synchronize(c5_context* ctx);

/* --- frames delivered to host memory, pipelined ----------------------------------------------------
 * plane::trace_rays hands back HOST pixels (plane.cpp:144-172); over PCIe Gen5 a 2400x1800 image is
 * 0.65 ms of transfer beside 0.7 ms of rendering, so the two are overlapped: c5_render_host_async renders
 * frame k into one of C5_HOST_RING internal device images and copies it to out_host on a copy stream of
 * its own while frame k + 1 is already being rendered.  out_host should be pinned (c5_host_alloc): a
 * pageable buffer makes the copy synchronous.  At most C5_HOST_RING frames may be outstanding; every
 * c5_render_host_async is paired, in order, with one c5_render_host_wait, which returns when THAT frame's
 * pixels are in out_host.  C5_RETRY from the wait: that frame and every frame enqueued after it are
 * incomplete (an internal buffer was too small and has been grown, or the grid turned out to need "algorithm" 1) —
 * wait for the rest, discard, render again.  A frame's status is read from ITS OWN counters (every frame of the ring
 * keeps its statistics apart: no frame in flight can add to another's).  Whichever call notices a failure first
 * (c5_get_stats, c5_get_row_costs, c5_synchronize between an async and its wait included) settles it and returns
 * C5_RETRY once; the waits of every frame outstanding at that moment return C5_RETRY as well, whatever their own
 * status words read; c5_last_error names the words. */
#define C5_HOST_RING 3
int c5_render_host_async(c5_context* ctx, float* out_host);
int c5_render_host_wait(c5_context* ctx);
/* The same, but frame_host is the FULL res_y x res_x image shared by all the contexts that render it
 * (c5_set_row_tiles / c5_set_row_range, one context per GPU): this context's rows are copied straight to
 * their final places in it — one strided copy over the context's own PCIe link, no exchange between GPUs and
 * no reassembly (the "N direct copies" of SURVEY.md section 8(e); with 8 GPUs, 8 links instead of the root's one).
 * Paired with c5_render_host_wait like c5_render_host_async. */
int c5_render_frame_rows_async(c5_context* ctx, float* frame_host);
/* Pinned host memory for images (hipHostMalloc with hipHostMallocPortable: several contexts — one per GPU — may copy
 * their rows into the same frame, c5_render_frame_rows_async / hipHostFree). */
int c5_host_alloc(c5_context* ctx, size_t bytes, void** out_ptr);
int c5_host_free(c5_context* ctx, void* ptr);

/* Statistics of the last completed frame (synchronizes). */
int c5_get_stats(c5_context* ctx, c5_stats* out);
/* Average duration (ms) of the walk kernel over the launches since the last call with
 * reset != 0; measured with HIP events on the context's stream. */
int c5_walk_kernel_ms(c5_context* ctx, int reset, double* avg_ms, int64_t* launches);

/* Host-only helper (no GPU needed): face adjacency of a conforming tetrahedral grid, the table
 * c5_upload_grid builds internally.  adj[n_cells][4] = neighbour across face f or -1, with the
 * reference's face numbering 0:(0,1,2) 1:(0,1,3) 2:(0,2,3) 3:(1,2,3) (plane.cpp:16-21).
 * Returns C5_ERR_MESH when a face is shared by more than two cells. */
int c5_face_adjacency(const int32_t* cell_vert, int64_t n_cells, int64_t n_pts, int32_t* adj,
                      int64_t* n_boundary_faces);

/* Host-only helper (no GPU needed): the point welding c5_upload_grid applies before it builds the
 * adjacency.  rep[i] = smallest point id whose coordinates equal point i's (rep[i] == i where nothing
 * coincides).  The reference copies four points per cell and ignores ids (object3d_base.cpp:37-42), so
 * coincident points are one point there by construction. */
int c5_weld_points(const double* xyz, int64_t n_pts, int32_t* rep, int64_t* n_merged);

/* Debug/inspection: transformed grid vertices of the last frame, xyz[n_pts][3]. */
int c5_download_view_points(c5_context* ctx, double* xyz);

#ifdef __cplusplus
}
#endif
#endif /* COURSE5_HIP_H */
