"""Host-side logic: synthetic grids, row-tile sharding, view rotation list.  CPU only."""
import os

import numpy as np
import pytest

from course5_amd import meshgen as mg, sharding


def test_cube8_is_the_documented_fixture():
    xyz, cells = mg.cube8()
    assert xyz.shape == (9, 3) and cells.shape == (8, 4)
    assert np.isclose(mg.signed_volumes(xyz, cells).sum(), 1.0)
    assert xyz[:, 0].min() == 0.5 and xyz[:, 0].max() == 1.5


@pytest.mark.parametrize("n", [1, 3, 5])
def test_kuhn_box_fills_the_cube_with_positive_cells(n):
    xyz, cells = mg.kuhn_box(n, jitter=0.1)
    assert cells.shape == (6 * n ** 3, 4) and xyz.shape == ((n + 1) ** 3, 3)
    vol = mg.signed_volumes(xyz, cells)
    assert (vol > 0).all() and np.isclose(vol.sum(), 1.0)


def test_c3_grid_has_the_surveyed_size():
    xyz, cells = mg.kuhn_box(55, jitter=0.1)
    assert cells.shape[0] == 998_250 and xyz.shape[0] == 175_616


def test_ball_is_non_convex_and_compacted():
    xyz, cells = mg.ball(12, 0.45)
    assert cells.max() == len(xyz) - 1 and len(np.unique(cells)) == len(xyz)
    assert (mg.signed_volumes(xyz, cells) > 0).all()


def test_validate_rejects_inverted_cells():
    xyz, cells = mg.cube8()
    cells = cells.copy()
    cells[0, [2, 3]] = cells[0, [3, 2]]
    with pytest.raises(ValueError):
        mg.validate(xyz, cells)


def test_view_rotations_follow_main_cpp():
    r = mg.view_rotations(0.5, 0.25, 0.1)
    pi = 3.14159265358979323846
    mp = -0.1 * pi + pi / 2
    assert r.tolist() == [[0.0, mp, 0.0], [1.0, 0.25 * pi, 1.0], [0.0, -mp + 0.5 * pi, 0.0]]


@pytest.mark.parametrize("res_y,tile_rows,world", [(1800, 16, 8), (450, 16, 3), (90, 7, 2), (10, 16, 4)])
def test_row_tiles_partition_the_image(res_y, tile_rows, world):
    seen = np.concatenate([sharding.local_rows(res_y, tile_rows, r, world) for r in range(world)])
    assert sorted(seen.tolist()) == list(range(res_y))
    strips = []
    full = np.arange(res_y * 5 * 2, dtype=np.float32).reshape(res_y, 5, 2)
    pad = sharding.padded_rows(res_y, tile_rows, world)
    for r in range(world):
        rows = sharding.local_rows(res_y, tile_rows, r, world)
        s = np.zeros((pad, 5, 2), dtype=np.float32)
        s[: rows.size] = full[rows]
        strips.append(s)
    assert np.array_equal(sharding.assemble(strips, res_y, tile_rows, world), full)


def test_vtk_writer_round_trip(tmp_path):
    xyz, cells, a, q = mg.workload("c1")
    p = tmp_path / "c1.vtk"
    mg.write_vtk_ascii(str(p), xyz, cells, a, q)
    text = p.read_text().split("\n")
    assert text[3] == "DATASET UNSTRUCTURED_GRID" and text[4] == "POINTS 9 double"
    assert "SCALARS AbsorpCoef double 1" in text and "SCALARS radEnLooseRate double 1" in text


def test_balanced_blocks_equalise_cost_and_cover_the_image():
    rng = np.random.default_rng(3)
    costs = np.zeros(1800)
    costs[400:1400] = rng.integers(50_000, 250_000, 1000)  # covered rows cluster in the middle
    for world in (1, 2, 3, 8):
        blocks = sharding.balanced_blocks(costs, world, base_cost=2400 * 0.05)
        assert blocks[0][0] == 0 and sum(n for _, n in blocks) == 1800
        assert all(blocks[k][0] + blocks[k][1] == blocks[k + 1][0] for k in range(world - 1))
        assert all(n >= 1 for _, n in blocks)
        per = [costs[b:b + n].sum() + n * 120 for b, n in blocks]
        assert max(per) <= 1.05 * (sum(per) / world)
    eq = sharding.equal_blocks(1800, 8)
    per_eq = [costs[b:b + n].sum() for b, n in eq]
    assert max(per_eq) > 1.5 * (sum(per_eq) / 8)  # equal blocks would be badly unbalanced


def test_blocks_cut_again_by_measured_times_even_out_what_the_model_misses():
    """sharding.time_weighted_costs: the ranks' times for the model's blocks rescale the rows' costs block by block.  A
    machine whose rows in the upper third cost 1.6 times what the model thinks, plus a fixed cost per share: two rounds
    of feedback bring the slowest share within 5 % of the mean (the scaling is per block: a change of rate inside a block is
    found only as the cuts move); the model's own blocks are 30 % off."""
    rng = np.random.default_rng(5)
    costs = np.zeros(1800)
    costs[300:1500] = rng.integers(80_000, 120_000, 1200)
    true_row = (costs + 2400 * 2.0) * np.where(np.arange(1800) < 600, 1.6, 1.0)

    def times(blocks):
        return [6.0e6 + true_row[b:b + n].sum() for b, n in blocks]  # (a quarter of a share's time is fixed)

    world = 8
    blocks = sharding.balanced_blocks(costs, world, base_cost=2400 * 3.0)
    t0 = times(blocks)
    assert max(t0) > 1.25 * (sum(t0) / world)
    for _ in range(2):
        blocks = sharding.balanced_blocks(sharding.time_weighted_costs(costs, blocks, times(blocks), base_cost=2400 * 3.0), world)
        assert blocks[0][0] == 0 and sum(n for _, n in blocks) == 1800
    t = times(blocks)
    assert max(t) < 1.05 * (sum(t) / world), (t0, t)
    # a rank without a time (0) keeps its model costs; nothing divides by zero on an empty block
    w = sharding.time_weighted_costs(costs, [(0, 900), (900, 900)], [0.0, 1.0], base_cost=1.0)
    assert np.array_equal(w[:900], costs[:900] + 1.0) and abs(w[900:].sum() - 1.0) < 1e-9


def test_balanced_blocks_degenerate_inputs():
    assert sharding.balanced_blocks(np.zeros(10), 4) == [(0, 1), (1, 1), (2, 1), (3, 7)] or \
        sum(n for _, n in sharding.balanced_blocks(np.zeros(10), 4)) == 10
    with pytest.raises(ValueError):
        sharding.balanced_blocks(np.ones(3), 4)
    full = np.arange(12 * 3 * 2, dtype=np.float32).reshape(12, 3, 2)
    blocks = sharding.balanced_blocks(np.arange(12), 3)
    strips = [np.vstack([full[b:b + n], np.zeros((2, 3, 2), np.float32)]) for b, n in blocks]
    assert np.array_equal(sharding.assemble_blocks(strips, blocks, 12), full)


def test_local_row_span_matches_brute_force(tmp_path):
    """device_types.hpp: local_row_span (the entry raster enumerates only a context's own rows with it)
    against local_row_of / global_row_of by brute force over worlds, ranks, tile heights and row ranges."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "row_span_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "course5_amd", "csrc"),
                    os.path.join(root, "tests", "cpp", "row_span_check.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert out.startswith("ok "), out
