"""BASELINE.json configs 3, 4 and 5 at their full sizes under `-m gpu`, each against the CPU oracle.

 C3  2400x1800 on the 998 250-cell grid: the whole frame against the oracle (which needs a few seconds
     and ~3 GB for it), not only against the second GPU implementation.
 C4  4800x3600 on the same grid, split into cyclic 16-row tiles over 8 shards (the row-tile split of
     config 4: each shard is what one of the 8 GPUs renders), rendered in turn on the one GPU of the test
     box, reassembled, compared bit for bit with the single-context frame and with the oracle on every
     64th row (oracle/oracle.cpp: Scene::keeps — pixels are independent, plane.cpp:161-169).
 C5  2400x1800, the 1M-cell grid with the Roche lobe and the sphere as `./course` always renders them
     (main.cpp:110-116,127), donor-angle steps (object3d_roche_lobe.cpp:48): the NaN mask of every frame
     against the oracle's solid raster, the volume pixels against the frame without solids.
"""
import numpy as np
import pytest

from course5_amd import capi, meshgen as mg, sharding
from parity import assert_images_match

pytestmark = pytest.mark.gpu
PI = 3.14159265358979323846


@pytest.fixture(scope="module")
def c3(gpu_ctx):
    xyz, cells, alpha, q = mg.workload("c3")
    for k in range(8):
        gpu_ctx.set_solid(k, np.zeros((0, 12)))
    for name, v in (("tile", 3), ("integration", 0), ("lds_stage", 2), ("algorithm", 0), ("xcd_mode", 2), ("depth_split", 0)):
        gpu_ctx.set_option(name, v)
    gpu_ctx.set_row_range(0, -1)
    gpu_ctx.set_row_tiles(0, 0, 1)
    gpu_ctx.upload_grid(xyz, cells, alpha, q)
    gpu_ctx.set_alpha_limit(2.5)
    gpu_ctx.set_view(mg.view_rotations(**mg.BENCH_VIEW))
    return xyz, cells, alpha, q


def test_c3_full_frame_against_the_cpu_oracle(gpu_ctx, oracle_port, c3):
    xyz, cells, alpha, q = c3
    rots = mg.view_rotations(**mg.BENCH_VIEW)
    gpu_ctx.set_image(2400, 1800, mg.REFERENCE_BOUNDS)
    img = gpu_ctx.render()
    st = gpu_ctx.stats()
    ref = oracle_port.render(xyz, cells, alpha, q, rots, 2400, 1800, mg.REFERENCE_BOUNDS, threads=16)
    assert st["segments"] == ref["segments"] == 170_283_916
    assert st["covered_pixels"] == ref["covered"]
    r = assert_images_match(img, ref["image"], "C3 at 2400x1800 vs the oracle")
    assert r["differing"] <= img.size // 1000, r


def test_c4_row_tile_split_at_4800x3600(gpu_ctx, oracle_port, c3):
    xyz, cells, alpha, q = c3
    rots = mg.view_rotations(**mg.BENCH_VIEW)
    rx, ry, world, tile_rows = 4800, 3600, 8, 16
    gpu_ctx.set_option("depth_split", 1)  # bit-equality across tilings: whole rays in every render
    gpu_ctx.set_image(rx, ry, mg.REFERENCE_BOUNDS)
    full = gpu_ctx.render()
    st_full = gpu_ctx.stats()
    strips, seg, cov = [], 0, 0
    for rank in range(world):
        gpu_ctx.set_row_tiles(tile_rows, rank, world)
        assert gpu_ctx.local_rows == sharding.local_row_count(ry, tile_rows, rank, world)
        strips.append(gpu_ctx.render())
        s = gpu_ctx.stats()
        seg += s["segments"]
        cov += s["covered_pixels"]
        # cyclic tiles balance by construction: every shard carries an eighth of the work within 2 %
        assert abs(s["segments"] - st_full["segments"] / world) < 0.02 * st_full["segments"] / world
    gpu_ctx.set_row_tiles(0, 0, 1)
    img = sharding.assemble(strips, ry, tile_rows, world)
    assert np.array_equal(img.view(np.uint32), full.view(np.uint32))
    assert seg == st_full["segments"] and cov == st_full["covered_pixels"]
    stride, phase = 64, 5
    ref = oracle_port.render(xyz, cells, alpha, q, rots, rx, ry, mg.REFERENCE_BOUNDS, threads=16,
                             row_stride=stride, row_phase=phase)
    assert ref["segments"] > 9_000_000
    assert_images_match(img[phase::stride], ref["image"][phase::stride], "C4 rows vs the oracle")
    # segments of exactly those rows (c5_get_row_costs counts them per row)
    gpu_ctx.set_option("row_costs", 1)
    gpu_ctx.render()
    costs = gpu_ctx.row_costs().astype(np.int64)
    gpu_ctx.set_option("row_costs", 0)
    assert int(costs[phase::stride].sum()) == ref["segments"] and int(costs.sum()) == st_full["segments"]


def test_c5_donor_sweep_with_lobe_and_sphere_on_the_1m_grid(gpu_ctx, oracle_port, c3, product_solids):
    xyz, cells, alpha, q = c3
    rots = mg.view_rotations(**mg.BENCH_VIEW)
    rx, ry = 2400, 1800
    gpu_ctx.set_image(rx, ry, mg.REFERENCE_BOUNDS)
    bare = gpu_ctx.render()
    lobe, sphere = product_solids
    gpu_ctx.set_solid(0, lobe.reshape(-1, 12), float("nan"))
    gpu_ctx.set_solid(1, sphere.reshape(-1, 12), float("nan"))
    gpu_ctx.set_solid_view(1, np.zeros((0, 3)))  # the sphere is never rotated (main.cpp:116)
    # the oracle only has to rasterise the solids: a one-cell grid far outside the view stands in for the volume
    tiny_xyz = np.array([[50.0, 50, 50], [51, 50, 50], [50, 51, 50], [50, 50, 51]])
    tiny_cells = np.array([[0, 1, 2, 3]], dtype=np.int32)
    masks = []
    try:
        for k in (0, 1, 45, 170):  # D = k / 180
            donor = k / 180.0
            lobe_rots = np.vstack([[1.0, donor * PI, 1.0], rots])  # object3d_roche_lobe.cpp:48, then main.cpp:112-114
            gpu_ctx.set_solid_view(0, lobe_rots)
            img = gpu_ctx.render()
            st = gpu_ctx.stats()
            lobe_view = oracle_port.rotate_points(lobe.reshape(-1, 3), lobe_rots).reshape(-1, 12)
            ref = oracle_port.render(tiny_xyz, tiny_cells, np.ones(1), np.ones(1), np.zeros((0, 3)), rx, ry,
                                     mg.REFERENCE_BOUNDS, solid_tets=np.vstack([lobe_view, sphere.reshape(-1, 12)]),
                                     solid_colour=float("nan"), threads=16)
            nan = np.isnan(img[..., 0])
            assert np.array_equal(nan, np.isnan(ref["image"][..., 0])), k   # NaN mask bit-exact (a9)
            assert np.array_equal(nan, np.isnan(img[..., 1]))
            assert st["solid_pixels"] == ref["marked"] == int(nan.sum()) > 50_000
            # -D turns only the lobe: every other pixel is the solids-free frame, to the bit
            assert np.array_equal(img[~nan].view(np.uint32), bare[~nan].view(np.uint32)), k
            assert st["segments"] < 170_283_916  # rays behind a solid are not walked
            masks.append(nan)
    finally:
        gpu_ctx.set_solid(0, np.zeros((0, 12)))
        gpu_ctx.set_solid(1, np.zeros((0, 12)))
    assert all(not np.array_equal(masks[0], m) for m in masks[1:])


def test_a_solids_own_mask_follows_every_change(gpu_ctx, product_solids):
    """Option "solid_cache" (default on): a solid whose view and image are those of the frame before is rastered once
    into a mask of its own and laid over later frames.  Every way that mask can go stale — the solid's view, the
    other solid's view, the image, the solid's geometry — is walked through, each state rendered several frames in a
    row (so that the first, the mask-building and the overlay frame all occur), and every frame must equal the one
    rendered with the cache switched off, bit for bit."""
    xyz, cells, alpha, q = mg.workload("g2")
    gpu_ctx.upload_grid(xyz, cells, alpha, q)
    rots = mg.view_rotations(0.1, 0.07)
    gpu_ctx.set_view(rots)
    lobe, sphere = product_solids
    lobe_rots = lambda d: np.vstack([[1.0, d * PI, 1.0], rots])  # noqa: E731

    def frames(n):
        out = []
        for _ in range(n):
            out.append(gpu_ctx.render().copy())
        return out

    def check(label):
        cached = frames(4)
        gpu_ctx.set_option("solid_cache", 0)
        plain = gpu_ctx.render()
        gpu_ctx.set_option("solid_cache", 1)
        for k, img in enumerate(cached):
            assert np.array_equal(img.view(np.uint32), plain.view(np.uint32)), (label, k)
        return plain

    try:
        gpu_ctx.set_image(600, 450, mg.REFERENCE_BOUNDS)
        gpu_ctx.set_solid(0, lobe.reshape(-1, 12), float("nan"))
        gpu_ctx.set_solid(1, sphere.reshape(-1, 12), 7.5)
        gpu_ctx.set_solid_view(0, lobe_rots(0.0))
        gpu_ctx.set_solid_view(1, np.zeros((0, 3)))
        a = check("both still")
        assert np.isnan(a[..., 0]).sum() > 1000 and (a[..., 0] == 7.5).sum() > 50
        gpu_ctx.set_solid_view(0, lobe_rots(0.3))            # the lobe moves, the sphere's own mask stays valid
        b = check("lobe moved")
        assert not np.array_equal(np.isnan(a[..., 0]), np.isnan(b[..., 0]))
        gpu_ctx.set_solid_view(1, np.array([[1.0, 0.2 * PI, 0.6]]))  # now the sphere moves (about an axis off its centre)
        c = check("sphere moved")
        assert not np.array_equal(c[..., 0] == 7.5, b[..., 0] == 7.5)
        gpu_ctx.set_image(480, 360, mg.REFERENCE_BOUNDS)     # another image: both masks stale
        check("image changed")
        gpu_ctx.set_image(600, 450, mg.REFERENCE_BOUNDS)
        bigger = (sphere.reshape(-1, 3) - np.array([1.0, 0, 0])) * 1.6 + np.array([1.0, 0, 0])
        gpu_ctx.set_solid(1, bigger.reshape(-1, 12), 7.5)    # other geometry in the same slot, same view
        d = check("geometry changed")
        assert not np.array_equal(d[..., 0] == 7.5, c[..., 0] == 7.5)
        gpu_ctx.set_row_tiles(16, 1, 2)                      # a shard of the image: masks are per local row
        check("sharded")
        gpu_ctx.set_row_tiles(16, 0, 1)
        check("whole again")
    finally:
        gpu_ctx.set_row_tiles(16, 0, 1)
        gpu_ctx.set_solid(0, np.zeros((0, 12)))
        gpu_ctx.set_solid(1, np.zeros((0, 12)))
