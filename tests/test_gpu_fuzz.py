"""Bounded randomised parity sweep on the GPU (tests/fuzz_parity.py holds the scene generator):
random grids with holes / disconnected parts / anisotropic scaling, random views, scalars, image
sizes and kernel variants, each compared with the CPU oracle."""
import importlib.util
import os

import numpy as np
import pytest

from course5_amd import meshgen as mg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tests", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("block", range(4))
def test_random_scenes_match_the_oracle(gpu_ctx, oracle_port, block):
    fuzz = _fuzz()
    scene = fuzz.scene
    gpu_ctx.set_row_tiles(0, 0, 1)
    gpu_ctx.set_row_range(0, -1)
    for k in range(10):
        seed = 1000 + 10 * block + k
        xyz, cells, alpha, q, rots, res, limit = scene(seed)
        try:
            ref = oracle_port.render(xyz, cells, alpha, q, rots, res[0], res[1], mg.REFERENCE_BOUNDS,
                                     alpha_limit=limit, threads=8)
        except RuntimeError:
            continue  # the reference algorithm itself aborts on exactly degenerate alignment
        gpu_ctx.upload_grid(xyz, cells, alpha, q)
        gpu_ctx.set_image(res[0], res[1], mg.REFERENCE_BOUNDS)
        gpu_ctx.set_view(rots)
        gpu_ctx.set_alpha_limit(limit)
        lds, order, tile = fuzz.VARIANTS[k % len(fuzz.VARIANTS)]
        gpu_ctx.set_option("lds_stage", min(lds, 2))
        gpu_ctx.set_option("stage_slots", 21 if lds == 3 else 14)
        gpu_ctx.set_option("integration", order)
        gpu_ctx.set_option("tile", tile)
        img = gpu_ctx.render()
        st = gpu_ctx.stats()
        a, b = img.astype(np.float64), ref["image"].astype(np.float64)
        tol = 1e-5 * np.maximum(np.abs(a), np.abs(b)) + 1e-6 * np.abs(b).max()
        assert int((np.abs(a - b) > tol).sum()) == 0, seed
        assert st["segments"] == ref["segments"] and st["covered_pixels"] == ref["covered"], seed
    gpu_ctx.set_option("lds_stage", 2)
    gpu_ctx.set_option("stage_slots", 0)
    gpu_ctx.set_option("integration", 0)
    gpu_ctx.set_option("tile", 3)
    gpu_ctx.set_alpha_limit(2.5)
