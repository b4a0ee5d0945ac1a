"""Grids that are conforming in space but not in connectivity: a coarse face against several fine
ones, single hanging nodes on edges.  The reference copies four points per cell and never looks at connectivity
(object3d_base.cpp:37-42), bins every face (plane.cpp:184-192) and sorts (line.cpp:138), so it renders them like
any other grid; here both faces of such an interface are boundary faces, and a ray that leaves through one has to
be picked up by the other at the same depth to rounding (walk_common.hpp: next_entry, entry_key_slack).

Golden fixtures g7 / g8 (the reference's own object code) run through test_golden_vectors* in every kernel variant;
this file adds fresh scenes against the CPU oracle, the unwelded variant, sharded renders, and the evidence that the
fixtures exercise the hole at all: with the entries keyed at their faces' own depth ("entry_key" 0, the behaviour
before round 3) the same frames lose rays."""
import numpy as np
import pytest

from course5_amd import meshgen as mg, sharding
from parity import assert_images_match

pytestmark = pytest.mark.gpu

VIEWS = ((0.1, 0.07), (0.3, 0.02), (0.0, 0.004), (0.37, -0.61), (-0.45, 0.93), (0.02, 0.5))


@pytest.fixture(autouse=True)
def _defaults(gpu_ctx):
    for k in range(8):
        gpu_ctx.set_solid(k, np.zeros((0, 12)))
    for name, v in (("tile", 3), ("integration", 0), ("lds_stage", 2), ("stage_slots", 0), ("algorithm", 0),
                    ("xcd_mode", 2), ("entry_key", 1), ("depth_split", 0)):
        gpu_ctx.set_option(name, v)
    gpu_ctx.set_row_tiles(0, 0, 1)
    gpu_ctx.set_row_range(0, -1)
    gpu_ctx.set_alpha_limit(2.5)
    yield
    gpu_ctx.set_option("entry_key", 1)


def _frame(ctx, rots, rx, ry):
    ctx.set_image(rx, ry, mg.REFERENCE_BOUNDS)
    ctx.set_view(rots)
    return ctx.render(), ctx.stats()


@pytest.mark.parametrize("weld", [True, False], ids=["shared-ids", "own-points-per-box"])
@pytest.mark.parametrize("warp", [0.0, 0.1], ids=["planar", "crumpled"])
def test_coarse_box_against_a_refined_one(gpu_ctx, oracle_port, warp, weld):
    xyz, cells, n_coarse = mg.refined_interface(4, 2, 4, jitter=0.12, warp=warp, seed=11, weld=weld)
    alpha, q = mg.scalars(len(cells), seed=12)
    gpu_ctx.upload_grid(xyz, cells, alpha, q)
    for view in VIEWS:
        rots = mg.view_rotations(*view)
        ref = oracle_port.render(xyz, cells, alpha, q, rots, 260, 190, mg.REFERENCE_BOUNDS, threads=8)
        for lds, order, tile in ((2, 0, 3), (2, 1, 3), (1, 0, 0), (0, 0, 1), (2, 0, 2)):
            gpu_ctx.set_option("lds_stage", lds)
            gpu_ctx.set_option("integration", order)
            gpu_ctx.set_option("tile", tile)
            img, st = _frame(gpu_ctx, rots, 260, 190)
            what = f"warp {warp} weld {weld} view {view} lds {lds} order {order} tile {tile}"
            assert st["segments"] == ref["segments"], what
            assert st["covered_pixels"] == ref["covered"], what
            assert_images_match(img, ref["image"], what)
            assert st["walk_overflow"] == 0
        if view == VIEWS[0]:  # the interface really is made of boundary faces: more entries than covered pixels
            assert st["entries"] > st["covered_pixels"]


def test_the_fixture_bites_without_the_depth_key(gpu_ctx, oracle_port):
    """"entry_key" 0 = entries keyed at their face's own depth, strict comparison (rounds 1-2): whether the abutting
    cell is found is then a coin toss per crossing, and rays were cut short with status C5_OK.  Since round 4 the walk
    notices: the abutting entry it could not take lies a rounding error BEFORE the depth at which the ray left the grid,
    i.e. inside the stretch the ray had walked - such rays are counted, the library answers C5_RETRY and renders the
    frame with bin_sort_resolve.  Either way the walk alone did not deliver the frame.  Keeps the fixtures honest."""
    xyz, cells, _ = mg.refined_interface(4, 2, 4, jitter=0.12, warp=0.1, seed=11)
    alpha, q = mg.scalars(len(cells), seed=12)
    gpu_ctx.upload_grid(xyz, cells, alpha, q)
    lost = 0
    for view in VIEWS[:4]:
        rots = mg.view_rotations(*view)
        ref = oracle_port.render(xyz, cells, alpha, q, rots, 260, 190, mg.REFERENCE_BOUNDS, threads=8)
        gpu_ctx.set_option("entry_key", 0)
        _, st0 = _frame(gpu_ctx, rots, 260, 190)
        gpu_ctx.set_option("entry_key", 1)
        _, st1 = _frame(gpu_ctx, rots, 260, 190)
        assert st1["segments"] == ref["segments"] and st1["steps"] > 0  # (setting "entry_key" lets the walk try again)
        assert st0["segments"] <= ref["segments"]
        lost += (ref["segments"] - st0["segments"]) + int(st0["steps"] == 0)  # rays cut short, or the frame redone without a walk
    assert lost > 0, "no ray was lost with the old keying: the fixture does not exercise the interface"


@pytest.mark.parametrize("seed", range(4))
def test_single_hanging_nodes(gpu_ctx, oracle_port, seed):
    rng = np.random.default_rng(500 + seed)
    n = int(rng.integers(3, 7))
    xyz, cells = mg.kuhn_box(n, lo=(0.6, -0.4, -0.4), size=0.8, jitter=0.12, seed=seed)
    for _ in range(int(rng.integers(1, 6))):
        cell = int(rng.integers(0, len(cells)))
        e = rng.choice(4, 2, replace=False)
        try:
            xyz, cells = mg.split_cell_at_edge_midpoint(xyz, cells, cell, (int(e[0]), int(e[1])))
        except ValueError:
            continue
    alpha = rng.uniform(0, 5, len(cells))
    q = rng.uniform(0, 2, len(cells))
    gpu_ctx.upload_grid(xyz, cells, alpha, q)
    for view in ((0.1, 0.07), (-0.62, 0.31), (0.45, -0.88)):
        rots = mg.view_rotations(*view)
        ref = oracle_port.render(xyz, cells, alpha, q, rots, 300, 220, mg.REFERENCE_BOUNDS, threads=8)
        for order in (0, 1):
            gpu_ctx.set_option("integration", order)
            img, st = _frame(gpu_ctx, rots, 300, 220)
            assert st["segments"] == ref["segments"] and st["covered_pixels"] == ref["covered"], (seed, view, order)
            assert_images_match(img, ref["image"], f"seed {seed} view {view} order {order}")


def test_refined_interface_orders_and_shards(gpu_ctx, oracle_port):
    xyz, cells, _ = mg.refined_interface(5, 2, 3, jitter=0.1, warp=0.06, seed=21)
    alpha, q = mg.scalars(len(cells), seed=22)
    gpu_ctx.upload_grid(xyz, cells, alpha, q)
    rx, ry = 320, 240
    for view in VIEWS[:4]:
        rots = mg.view_rotations(*view)
        ref = oracle_port.render(xyz, cells, alpha, q, rots, rx, ry, mg.REFERENCE_BOUNDS, threads=8)
        gpu_ctx.set_option("depth_split", 1)  # (the shards below must equal the full frame bit for bit)
        full, st = _frame(gpu_ctx, rots, rx, ry)
        assert st["segments"] == ref["segments"]
        assert_images_match(full, ref["image"], f"view {view}")
        # front to back: the same cells in the other order
        gpu_ctx.set_option("integration", 1)
        img, st1 = _frame(gpu_ctx, rots, rx, ry)
        assert_images_match(img, ref["image"], f"front to back, view {view}")
        assert st1["segments"] == ref["segments"]
        gpu_ctx.set_option("integration", 0)
        # sharded like the ranks of a multi-GPU run: cyclic tiles and blocks reassemble to the same bits
        strips = []
        for r in range(3):
            gpu_ctx.set_row_tiles(8, r, 3)
            strips.append(gpu_ctx.render())
        gpu_ctx.set_row_tiles(0, 0, 1)
        assert np.array_equal(sharding.assemble(strips, ry, 8, 3).view(np.uint32), full.view(np.uint32))


def test_refined_interface_at_scale(gpu_ctx, oracle_port):
    """A larger interface (13 824 + 110 592 cells, 1 152 coarse against 4 608 fine interface faces), at an image
    size where the oracle still finishes in seconds: every pixel, S and covered."""
    xyz, cells, _ = mg.refined_interface(24, 4, 8, lo=(0.55, -0.45, -0.45), size=0.9, jitter=0.1, warp=0.05, seed=31)
    alpha, q = mg.scalars(len(cells), seed=32)
    gpu_ctx.upload_grid(xyz, cells, alpha, q)
    rots = mg.view_rotations(0.1, 0.07)
    ref = oracle_port.render(xyz, cells, alpha, q, rots, 800, 600, mg.REFERENCE_BOUNDS, threads=8)
    img, st = _frame(gpu_ctx, rots, 800, 600)
    assert st["segments"] == ref["segments"] and st["covered_pixels"] == ref["covered"]
    r = assert_images_match(img, ref["image"], "refined interface 24")
    assert r["differing"] <= img.size // 1000, r


@pytest.mark.parametrize("view", [(0.1, 0.07), (0.37, -0.61)], ids=["bench-view", "oblique"])
def test_refined_interface_at_the_benchmark_size(gpu_ctx, oracle_port, view):
    """BASELINE's own scale: 98 304 coarse + 786 432 fine cells (884 736; 2 048 coarse interface faces against
    8 192 fine ones) at 2400x1800 — every pixel of the frame, S and covered against the CPU oracle, in both
    integration orders, plus an 8-way cyclic split reassembled bit for bit.  In the bench view the
    interface is seen at 13 degrees (5 % of the rays cross it), in the oblique one most rays do."""
    xyz, cells, n_coarse = mg.refined_interface(32, 16, 32, lo=(0.5, -0.5, -0.5), size=1.0, jitter=0.1, warp=0.05, seed=77)
    assert n_coarse == 98_304 and len(cells) == 884_736
    alpha, q = mg.scalars(len(cells), seed=78)
    rots = mg.view_rotations(*view)
    gpu_ctx.upload_grid(xyz, cells, alpha, q)
    ref = oracle_port.render(xyz, cells, alpha, q, rots, 2400, 1800, mg.REFERENCE_BOUNDS, threads=16)
    gpu_ctx.set_option("depth_split", 1)  # (the shards below must equal the full frame bit for bit: whole rays everywhere)
    full, st = _frame(gpu_ctx, rots, 2400, 1800)
    assert st["segments"] == ref["segments"] and st["covered_pixels"] == ref["covered"]
    assert st["entries"] > st["covered_pixels"] * (1.03 if view == (0.1, 0.07) else 1.4)  # rays that cross the interface enter twice
    r = assert_images_match(full, ref["image"], f"refined interface, 884 736 cells, 2400x1800, view {view}")
    assert r["differing"] <= full.size // 1000, r
    gpu_ctx.set_option("integration", 1)
    img, st1 = _frame(gpu_ctx, rots, 2400, 1800)
    assert st1["segments"] == ref["segments"]
    assert_images_match(img, ref["image"], "front to back")
    gpu_ctx.set_option("integration", 0)
    strips = []
    for rank in range(8):
        gpu_ctx.set_row_tiles(16, rank, 8)
        strips.append(gpu_ctx.render())
    gpu_ctx.set_row_tiles(0, 0, 1)
    assert np.array_equal(sharding.assemble(strips, 1800, 16, 8).view(np.uint32), full.view(np.uint32))
