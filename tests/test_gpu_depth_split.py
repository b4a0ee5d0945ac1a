"""Option "depth_split" (round 4): a ray cut at K - 1 planes of constant depth, the K parts walked by K
jobs per 8x8 tile, the partial integrals composed in depth order (tau = sum tau_s; I <- exp(-tauc_s) I + b_s: the
recurrence of line.cpp:206-225 is affine in I).  Same bar as every parity test: 1e-5 relative against the golden vectors
made by the reference's object code and against the CPU oracle, segment and covered-pixel counts EQUAL (a cell cut by a
plane is counted by the job in which the ray leaves it).  What a split changes is the rounding order of I (~1e-16), which
is why it is never chosen by itself for a grid with a cell whose clamped alpha lies in [DBL_EPSILON, 1e-6): there the
reference's own result is cancellation noise of the very bits of I (fixture g4)."""
import os

import numpy as np
import pytest

from course5_amd import capi, meshgen as mg, sharding
from parity import assert_images_match, compare, golden_fixtures, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _defaults(gpu_ctx):
    for k in range(8):
        gpu_ctx.set_solid(k, np.zeros((0, 12)))
    for name, v in (("tile", 3), ("integration", 0), ("lds_stage", 2), ("stage_slots", 0), ("algorithm", 0), ("xcd_mode", 2),
                    ("entry_key", 1), ("view_cache", 1), ("depth_split", 0), ("split_tilt_x", 0.0), ("split_tilt_y", 0.0)):
        gpu_ctx.set_option(name, v)
    gpu_ctx.set_row_tiles(0, 0, 1)
    gpu_ctx.set_row_range(0, -1)
    gpu_ctx.set_alpha_limit(2.5)
    yield
    gpu_ctx.set_option("depth_split", 0)
    gpu_ctx.set_option("stage_slots", 0)
    gpu_ctx.set_option("view_cache", 1)
    gpu_ctx.set_option("split_tilt_x", 0.0)
    gpu_ctx.set_option("split_tilt_y", 0.0)


def _frame(ctx, rots, rx, ry, bounds=mg.REFERENCE_BOUNDS):
    ctx.set_image(rx, ry, bounds)
    ctx.set_view(rots)
    return ctx.render(), ctx.stats()


@pytest.mark.parametrize("tilt", [(0.0, 0.0), (0.31, -0.17), (-1.3, 0.8)], ids=lambda t: f"tilt{t[0]}_{t[1]}")
@pytest.mark.parametrize("slabs", [2, 3, 4, 7])
@pytest.mark.parametrize("path", golden_fixtures(), ids=lambda p: os.path.basename(p)[:-4])
def test_golden_vectors_with_rays_cut_in_slabs(gpu_ctx, path, slabs, tilt):
    """Every golden fixture (G1, G2, G7 / G8 with hanging nodes, the non-convex ball) with the split forced: images
    within the bar of the reference's, S and covered pixels equal; both slot counts of the walk.  The cutting planes
    of constant depth, and tilted (depth - gx x - gy y constant: what the library fits by itself to an oblique view)."""
    gpu_ctx.set_option("split_tilt_x", tilt[0])
    gpu_ctx.set_option("split_tilt_y", tilt[1])
    fx = load_golden(path)
    if fx["name"].startswith("g4_"):
        pytest.skip("g4 holds alpha in [DBL_EPSILON, 1e-8): the reference's recurrence is its own cancellation noise there; "
                    "the library never splits such a grid by itself (test_g4_is_walked_whole_unless_forced)")
    rx, ry = (int(v) for v in fx["res"])
    stride = int(fx["stride"])
    gpu_ctx.upload_grid(fx["xyz"], fx["cells"], fx["alpha"], fx["q"])
    gpu_ctx.set_alpha_limit(float(fx["alpha_limit"]))
    for k in range(len(fx["views"])):
        for slots in (14, 21):
            gpu_ctx.set_option("stage_slots", slots)
            gpu_ctx.set_option("depth_split", 1)
            _, whole = _frame(gpu_ctx, fx[f"rots{k}"], rx, ry, fx["bounds"])
            gpu_ctx.set_option("depth_split", slabs)
            img, st = _frame(gpu_ctx, fx[f"rots{k}"], rx, ry, fx["bounds"])
            what = f"{fx['name']} view {k} slabs {slabs} slots {slots} tilt {tilt}"
            r = assert_images_match(img[::stride, ::stride], fx[f"image{k}"], what)
            assert st["segments"] == int(fx[f"segments{k}"]), what
            assert st["covered_pixels"] == int(fx[f"covered{k}"]), what
            assert st["walk_overflow"] == 0
            assert st["steps"] >= whole["steps"], what  # (a cell cut by a plane is stepped through by two jobs)
            assert r["max_rel"] < 1e-6, (what, r)


def test_g4_is_walked_whole_unless_forced(gpu_ctx):
    """Fixture g4 (alpha = 2.3e-16, 1e-9, ... : every branch of line.cpp:213-224): its image is pinned bit for bit by the
    whole-ray walk.  "depth_split" 0 must leave such a grid alone however small the frame (alpha floor below 1e-6); forced,
    the split is allowed to differ - and does, which is what the rule is for.  States the size of the difference."""
    fx = load_golden([p for p in golden_fixtures() if "g4_" in p][0])
    rx, ry = (int(v) for v in fx["res"])
    gpu_ctx.upload_grid(fx["xyz"], fx["cells"], fx["alpha"], fx["q"])
    gpu_ctx.set_alpha_limit(float(fx["alpha_limit"]))
    rots = fx["rots1"]
    gpu_ctx.set_option("depth_split", 1)
    whole, st_whole = _frame(gpu_ctx, rots, rx, ry, fx["bounds"])
    gpu_ctx.set_option("depth_split", 0)
    for _ in range(3):  # (the rule looks at the frame before)
        auto, st_auto = _frame(gpu_ctx, rots, rx, ry, fx["bounds"])
    assert np.array_equal(auto.view(np.uint32), whole.view(np.uint32)) and st_auto["steps"] == st_whole["steps"]
    assert_images_match(auto, fx["image1"], "g4, whole rays")
    gpu_ctx.set_option("depth_split", 3)
    forced, st_forced = _frame(gpu_ctx, rots, rx, ry, fx["bounds"])
    assert st_forced["segments"] == st_whole["segments"] and st_forced["steps"] > st_whole["steps"]
    r = compare(forced, fx["image1"])
    print(f"g4 with the split forced: {r}")
    assert_images_match(forced[..., :1], fx["image1"][..., :1], "g4 forced: tau does not depend on the order")


@pytest.mark.parametrize("view", [(0.1, 0.07), (0.9, -0.3), (1.0, 0.02), (0.35, -0.6)], ids=lambda v: f"X{v[0]}Y{v[1]}")
def test_nonconvex_ball_and_stacked_slabs_against_the_oracle(gpu_ctx, oracle_port, view):
    """Rays that leave and re-enter the grid between and across the planes: the C2 ball (staircase boundary) and four slabs
    with gaps one behind the other (a plane inside a gap: the job above starts from the next boundary entry)."""
    rots = mg.view_rotations(*view)
    xyz, cells, alpha, q = mg.workload("c2")
    gpu_ctx.upload_grid(xyz, cells, alpha, q)
    ref = oracle_port.render(xyz, cells, alpha, q, rots, 500, 375, mg.REFERENCE_BOUNDS, threads=8)
    for slabs in (2, 4, 5):
        gpu_ctx.set_option("depth_split", slabs)
        img, st = _frame(gpu_ctx, rots, 500, 375)
        assert st["segments"] == ref["segments"] and st["covered_pixels"] == ref["covered"], (view, slabs)
        assert_images_match(img, ref["image"], f"ball, view {view}, {slabs} slabs")
    pts, cls = [], []
    for k in range(4):
        x, c = mg.kuhn_box(3, lo=(0.6, -0.4, -0.45 + 0.24 * k), size=0.8, jitter=0.1, seed=5 + k)
        x = x.copy()
        x[:, 2] = -0.45 + 0.24 * k + (x[:, 2] - x[:, 2].min()) * 0.2
        cls.append(c + sum(len(p) for p in pts))
        pts.append(x)
    xyz, cells = np.vstack(pts), np.vstack(cls).astype(np.int32)
    cells = mg.orient_positive(xyz, cells)
    alpha, q = mg.scalars(len(cells), seed=5)
    gpu_ctx.upload_grid(xyz, cells, alpha, q)
    ref = oracle_port.render(xyz, cells, alpha, q, rots, 300, 220, mg.REFERENCE_BOUNDS, threads=8)
    for slabs in (2, 3, 4, 8):
        gpu_ctx.set_option("depth_split", slabs)
        img, st = _frame(gpu_ctx, rots, 300, 220)
        assert st["segments"] == ref["segments"] and st["covered_pixels"] == ref["covered"], (view, slabs)
        assert_images_match(img, ref["image"], f"stacked slabs, view {view}, {slabs} slabs")


def test_random_scenes_cut_in_slabs(gpu_ctx, oracle_port):
    """The scenes of the randomised sweep (holes, disconnected parts, hanging nodes, sub-pixel cells, alpha = 0 cells and
    cells above the clamp) with a random number of slabs."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(__file__), "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    done = 0
    for seed in range(7000, 7030):
        xyz, cells, alpha, q, rots, res, limit = fz.scene(seed)
        try:
            ref = oracle_port.render(xyz, cells, alpha, q, rots, res[0], res[1], mg.REFERENCE_BOUNDS, alpha_limit=limit, threads=8)
        except RuntimeError:
            continue
        gpu_ctx.upload_grid(xyz, cells, alpha, q)
        gpu_ctx.set_alpha_limit(limit)
        slabs = 2 + seed % 5
        gpu_ctx.set_option("depth_split", slabs)
        rng = np.random.default_rng(seed)
        tilt = rng.uniform(-1.5, 1.5, 2) if seed % 3 else np.zeros(2)
        gpu_ctx.set_option("split_tilt_x", float(tilt[0]))
        gpu_ctx.set_option("split_tilt_y", float(tilt[1]))
        img, st = _frame(gpu_ctx, rots, res[0], res[1])
        assert st["segments"] == ref["segments"] and st["covered_pixels"] == ref["covered"], (seed, slabs)
        assert_images_match(img, ref["image"], f"seed {seed}, {slabs} slabs")
        done += 1
    assert done >= 25


def test_shards_solids_and_a_view_that_stands_still(gpu_ctx, oracle_port):
    """The split under everything else a frame can be: rows sharded like the ranks of a multi-GPU run (cyclic tiles and a
    block), solids laid over it, and the view cache (per-view data, plane cells included, reused from the third frame)."""
    xyz, cells, alpha, q = mg.workload("c2")
    rots = mg.view_rotations(0.1, 0.07)
    gpu_ctx.upload_grid(xyz, cells, alpha, q)
    rx, ry = 480, 360
    ref = oracle_port.render(xyz, cells, alpha, q, rots, rx, ry, mg.REFERENCE_BOUNDS, threads=8)
    gpu_ctx.set_option("depth_split", 3)
    full, st = _frame(gpu_ctx, rots, rx, ry)
    assert st["segments"] == ref["segments"]
    assert_images_match(full, ref["image"], "full frame")
    for k in range(4):  # the same view again and again: built twice, then reused
        again = gpu_ctx.render()
        st_k = gpu_ctx.stats()
        assert np.array_equal(again.view(np.uint32), full.view(np.uint32)), k
        assert st_k["segments"] == st["segments"] and st_k["steps"] == st["steps"]
    assert st_k["ms_records"] == 0.0 and st_k["ms_entries"] == 0.0  # (reused: the three setup launches did not run)
    strips = []
    for r in range(3):
        gpu_ctx.set_row_tiles(16, r, 3)
        strips.append(gpu_ctx.render())
    gpu_ctx.set_row_tiles(0, 0, 1)
    assert np.array_equal(sharding.assemble(strips, ry, 16, 3).view(np.uint32), full.view(np.uint32))
    gpu_ctx.set_row_range(101, 77)
    block = gpu_ctx.render()
    gpu_ctx.set_row_range(0, -1)
    assert np.array_equal(block.view(np.uint32), full[101:178].view(np.uint32))
    s0x, s0c = mg.kuhn_box(2, lo=(0.9, -0.2, -0.2), size=0.3)
    gpu_ctx.set_solid(0, s0x[s0c], float("nan"))
    gpu_ctx.set_solid_view(0, rots)
    masked, st_m = _frame(gpu_ctx, rots, rx, ry)
    gpu_ctx.set_option("depth_split", 1)
    masked_whole, st_w = _frame(gpu_ctx, rots, rx, ry)
    assert st_m["solid_pixels"] == st_w["solid_pixels"] > 0 and st_m["segments"] == st_w["segments"]
    assert np.array_equal(np.isnan(masked), np.isnan(masked_whole))
    assert_images_match(masked, masked_whole, "solids over a split frame")


def test_small_frames_are_split_by_themselves_and_large_ones_are_not(gpu_ctx, oracle_port):
    """"depth_split" 0 (the default): a frame whose rays do not fill the GPU's wavefront slots and are long enough is cut
    from the next frame on (the rule looks at the statistics of the frame before) - the C3 grid at 800x600: 2 - 4 slabs;
    the same grid at 2400x1800 (2.5 rounds of jobs) stays whole.  Either way the image is the whole-ray image to the bar."""
    xyz, cells, alpha, q = mg.workload("c3")
    rots = mg.view_rotations(0.1, 0.07)
    gpu_ctx.upload_grid(xyz, cells, alpha, q)
    gpu_ctx.set_option("view_cache", 0)
    for res, expect_split in (((800, 600), True), ((2400, 1800), False)):
        gpu_ctx.set_option("depth_split", 1)
        whole, st_whole = _frame(gpu_ctx, rots, *res)
        gpu_ctx.set_option("depth_split", 0)
        for _ in range(3):
            img, st = _frame(gpu_ctx, rots, *res)
        assert st["segments"] == st_whole["segments"] and st["covered_pixels"] == st_whole["covered_pixels"]
        assert (st["steps"] > st_whole["steps"]) == expect_split, (res, st["steps"], st_whole["steps"])
        assert_images_match(img, whole, f"C3 at {res}")
        if expect_split:
            r = compare(img, whole)
            assert r["max_rel"] < 1e-9, r
    # one cell with alpha = 1e-9 (the reference's recurrence is cancellation noise there): such a grid is left whole
    alpha_ill = alpha.copy()
    alpha_ill[len(alpha_ill) // 2] = 1e-9
    gpu_ctx.update_scalars(alpha_ill, q)
    gpu_ctx.set_option("depth_split", 1)
    whole, st_whole = _frame(gpu_ctx, rots, 800, 600)
    gpu_ctx.set_option("depth_split", 0)
    for _ in range(3):
        img, st = _frame(gpu_ctx, rots, 800, 600)
    assert st["steps"] == st_whole["steps"] and np.array_equal(img.view(np.uint32), whole.view(np.uint32))
    gpu_ctx.update_scalars(alpha, q)
    ref = oracle_port.render(xyz, cells, alpha, q, rots, 800, 600, mg.REFERENCE_BOUNDS, threads=16)
    gpu_ctx.set_option("depth_split", 4)
    img, st = _frame(gpu_ctx, rots, 800, 600)
    assert st["segments"] == ref["segments"] and st["covered_pixels"] == ref["covered"]
    assert_images_match(img, ref["image"], "C3 at 800x600, 4 slabs, against the oracle")


def test_interpenetrating_boxes_are_still_noticed_when_rays_are_cut(gpu_ctx, oracle_port):
    """The entries a ray had to skip are judged per job (next_entry: also when a job ends at its plane): two overlapping
    boxes rendered with the split forced still come out of bin_sort_resolve."""
    xa, ca = mg.kuhn_box(3, lo=(0.6, -0.4, -0.3), size=0.6, jitter=0.1, seed=5)
    xb, cb = mg.kuhn_box(4, lo=(0.85, -0.2, -0.45), size=0.7, jitter=0.1, seed=6)
    xyz = np.vstack([xa, xb])
    cells = np.vstack([ca, cb + len(xa)]).astype(np.int32)
    alpha, q = mg.scalars(len(cells), seed=9)
    rots = mg.view_rotations(0.13, 0.21)
    gpu_ctx.upload_grid(xyz, cells, alpha, q)
    gpu_ctx.set_option("depth_split", 4)
    img, st = _frame(gpu_ctx, rots, 240, 180)
    ref = oracle_port.render(xyz, cells, alpha, q, rots, 240, 180, mg.REFERENCE_BOUNDS, threads=8)
    assert st["steps"] == 0 and st["segments"] == ref["segments"]
    assert_images_match(img[..., :1], ref["image"][..., :1], "overlapping boxes, tau")
