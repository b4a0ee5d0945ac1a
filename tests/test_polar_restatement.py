"""f-1 (procedural solids): the product's C++ restatement of init_polar / the Roche potential / the sphere
(course5_amd/csrc/host/scene.cpp) against a second restatement written independently in numpy from the
reference text (tests/polar_restatement.py), point for point.  CPU only.

The reference's own object code is out of reach for this part (object3d_base.cpp needs VTK headers), so geometry
parity of the solids stays "partial": two restatements by different routes agreeing BIT FOR BIT on all 652 802
cells is the pin that is possible here.  It is a sharp one: the sphere's 261 121 rays all end within rounding of
the level set (20 steps of 0.001 against R = 0.02), so a single differing last bit of a step vector moves a
point by a whole marching step (a handful do) — which is how this test found that g++ -O3 merges cos(a) and sin(a) into glibc's
sincos(), whose results differ from cos() / sin() for one angle in a thousand."""
import numpy as np

import polar_restatement as pr


def test_roche_lobe_and_sphere_match_the_numpy_restatement_bit_for_bit(product_solids):
    lobe, sphere = product_solids
    want_lobe, ring, top, bottom = pr.roche_lobe()
    assert ring.shape == (255, 256, 3)             # accumulated angles give 255 x 256 points
    assert want_lobe.shape == lobe.shape == (130_560, 4, 3)
    assert np.array_equal(want_lobe.view(np.uint64), lobe.view(np.uint64))
    want_sphere, ring, top, bottom = pr.sphere()
    assert ring.shape == (511, 511, 3)
    assert want_sphere.shape == sphere.shape == (522_242, 4, 3)
    assert np.array_equal(want_sphere.view(np.uint64), sphere.view(np.uint64))
    # the closing cell of the top fan uses ring 0, not the last ring (object3d_base.cpp:171-174)
    n = ring.shape[1]
    assert np.array_equal(sphere[2 * n - 1][3], ring[0][n - 1]) and not np.array_equal(ring[0][n - 1], ring[-1][n - 1])


def test_the_sphere_sits_on_a_knife_edge():
    """Why bit-equality is the only meaningful comparison here: with separate cos() / sin() calls instead of
    sincos() (1 ulp apart for ~1 angle in 1000) hundreds of sphere points land one marching step away."""
    import math
    saved = pr._sincos
    try:
        pr._sincos = lambda a: (math.sin(a), math.cos(a))
        other, _, _, _ = pr.sphere()
    finally:
        pr._sincos = saved
    ours, _, _, _ = pr.sphere()
    differs = (other != ours).any(axis=(1, 2))
    moved = np.abs(other - ours).max(axis=(1, 2)) > 5e-4  # by a whole marching step
    assert int(differs.sum()) > 50 and 1 <= int(moved.sum()) < 20_000 and np.abs(other - ours).max() < 1.1e-3
