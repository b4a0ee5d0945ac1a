"""Unit vectors for the coverage rule (a3-a7: plane.cpp:57-142,194-212 as restated in oracle/scan.hpp — the part
of the oracle no compiled reference code backs, because plane.cpp needs VTK headers).  Every expectation here is
derived WITHOUT the scanline code: closed point-in-triangle tests in exact rational arithmetic for the regular
branches, and pixel lists worked out by hand for what the reference does beyond plain coverage (a flat edge on a
pixel row, the clamp of plane::get_pixel_by_x/_y).  The grid is 10 x 8 pixels over [0, 9] x [0, 7], so pixel
(i, j) is the point (i, j) exactly and every hand computation is integer arithmetic.  CPU only."""
from fractions import Fraction as F

import numpy as np
import pytest

BOUNDS = (9.0, 0.0, 7.0, 0.0)  # {x_max, x_min, y_max, y_min}: 10 x 8 pixels, step 1
RX, RY = 10, 8


def covered(oracle, tri):
    px, n = oracle.scan_face(RX, RY, BOUNDS, *[np.array([x, y, 0.0]) for x, y in tri])
    assert n == len(px)
    return sorted(map(tuple, px.tolist())), px


def inside_closed(tri):
    """Pixel centres inside or on the border of the triangle, by exact edge functions."""
    (ax, ay), (bx, by), (cx, cy) = [(F(x), F(y)) for x, y in tri]
    out = []
    for j in range(RY):
        for i in range(RX):
            e = [(bx - ax) * (j - ay) - (by - ay) * (i - ax),
                 (cx - bx) * (j - by) - (cy - by) * (i - bx),
                 (ax - cx) * (j - cy) - (ay - cy) * (i - cx)]
            if all(v >= 0 for v in e) or all(v <= 0 for v in e):
                out.append((i, j))
    return sorted(out)


# coordinates are multiples of 1/8 (exact in binary), none on a pixel row / column unless the case says so
REGULAR = {
    # plane.cpp:66-89: which side of the long edge p0-p2 the middle vertex lies on, all four sign combinations
    "middle vertex right of an ascending long edge": [(1.25, 0.5), (6.75, 3.5), (2.5, 6.375)],
    "middle vertex left of an ascending long edge": [(1.25, 0.5), (0.25, 3.5), (2.5, 6.375)],
    "middle vertex right of a descending long edge": [(6.5, 0.625), (8.25, 2.5), (3.125, 6.25)],
    "middle vertex left of a descending long edge": [(6.5, 0.625), (1.125, 4.75), (3.125, 6.25)],
    "flat top between pixel rows": [(2.125, 5.5), (7.375, 5.5), (4.25, 1.25)],
    "flat bottom between pixel rows": [(2.125, 1.5), (7.375, 1.5), (4.25, 6.25)],
    "thinner than a pixel (may cover nothing)": [(3.25, 1.125), (3.375, 1.125), (3.3125, 6.5)],
    "sliver across the grid": [(0.25, 0.25), (8.75, 6.625), (7.5, 6.75)],
    "a vertex on a pixel centre, two edges through pixel centres": [(2.0, 1.0), (6.0, 1.5), (2.0, 5.0)],
}


@pytest.mark.parametrize("name", list(REGULAR))
def test_scan_conversion_is_the_closed_triangle(oracle_port, name):
    tri = REGULAR[name]
    want = inside_closed(tri)
    for order in ((0, 1, 2), (1, 2, 0), (2, 0, 1), (0, 2, 1)):  # the rule must not depend on the vertex order
        got, px = covered(oracle_port, [tri[k] for k in order])
        assert got == want, (name, order)
        assert len(set(got)) == len(got)
        # emission order: rows ascending, columns ascending within a row (plane.cpp:104-138)
        assert px.tolist() == sorted(px.tolist(), key=lambda p: (p[1], p[0]))
    if name.startswith("thinner"):
        assert want == []
    else:
        assert len(want) >= 3


def test_a_flat_edge_exactly_on_a_pixel_row(oracle_port):
    """plane::line_rev_function_eq returns p1.x for a horizontal edge (plane.cpp:50-55: abs(dy) < DBL_EPSILON), and
    on the row of a flat TOP edge the short edge is (p0, p1) — the flat one: the span collapses to what lies
    between the long edge and ONE end of the flat edge.  By hand for the triangle (2,5) (7,5) (4,1):
      sorted by y descending p0 = (2,5) or (7,5) (std::sort, equal keys), p2 = (4,1);
      rows 1..5;  row 5: y = 5 is not < p1.y -> short edge (p0,p1) horizontal -> x = p0.x; long edge at y=5: p0.x
      -> exactly one pixel, (2,5) or (7,5); the rows below are the plain closed triangle."""
    tri = [(2.0, 5.0), (7.0, 5.0), (4.0, 1.0)]
    got, _ = covered(oracle_port, tri)
    below = [p for p in inside_closed(tri) if p[1] < 5]
    assert [p for p in got if p[1] < 5] == below
    top = [p for p in got if p[1] == 5]
    assert top in ([(2, 5)], [(7, 5)])
    # a flat BOTTOM edge on a row: there y < p1.y never holds on that row either, but the short edge is then
    # (p2, p1) only for rows BELOW p1 — the bottom row itself uses (p0, p1) against the long edge (p0, p2): the span
    # between the two slanted edges at y = y_min, i.e. the whole bottom edge, like the closed triangle
    tri = [(2.0, 1.0), (7.0, 1.0), (4.0, 5.0)]
    got, _ = covered(oracle_port, tri)
    assert [p for p in got if p[1] > 1] == [p for p in inside_closed(tri) if p[1] > 1]
    bottom = [p for p in got if p[1] == 1]
    assert bottom in ([(i, 1) for i in range(2, 8)], [(2, 1)], [(7, 1)])


def test_geometry_beyond_the_domain_is_clamped_onto_the_border(oracle_port):
    """plane::get_pixel_by_x/_y clamp the fractional index to [0, N-1] (plane.cpp:194-212): a row span that lies
    wholly beyond x_max still yields ceil(9) .. floor(9) = column 9 — the "smear" of readme.md:46.  By hand:
    triangle (7.5, 1.5) (12.5, 1.5) (12.5, 5.5): hypotenuse x = 7.5 + 1.25 (y - 1.5), right edge x = 12.5.
      row 2: span [8.125, 12.5] -> ceil 9 (8.125 -> 9) .. floor(clamp 12.5 = 9) -> (9,2)
      row 3: [9.375, 12.5]: clamp(9.375) = 9 -> (9,3), although x = 9 is OUTSIDE the triangle there
      rows 4, 5: [10.625, ..], [11.875, ..] -> both ends clamp to 9 -> (9,4), (9,5)"""
    got, _ = covered(oracle_port, [(7.5, 1.5), (12.5, 1.5), (12.5, 5.5)])
    assert got == [(9, 2), (9, 3), (9, 4), (9, 5)]
    assert inside_closed([(7.5, 1.5), (12.5, 1.5), (12.5, 5.5)]) == [(9, 2)]  # geometrically only this one
    # wholly right of the domain: every row of its y range lands on column 9
    got, _ = covered(oracle_port, [(10.5, 1.5), (13.0, 2.5), (11.0, 4.5)])
    assert got == [(9, 2), (9, 3), (9, 4)]
    # left of x_min: column 0
    got, _ = covered(oracle_port, [(-3.5, 2.5), (-1.5, 2.75), (-2.0, 4.25)])
    assert got == [(0, 3), (0, 4)]
    # rows: a triangle above y_max.  ceil(clamp) .. floor(clamp) = row 7 only, scanned at y = Y[7] = 7 with the edge
    # functions extrapolated: triangle (2.5, 8.5) (6.5, 8.5) (4.5, 10.5): sorted p0 = (4.5,10.5), p1/p2 the two
    # bottom vertices; at y = 7 < p1.y the short edge is (p2, p1), horizontal -> x = p2.x; the long edge (p0, p2)
    # extrapolated to y = 7 gives x = p2.x -+ 1.5.  So the span is [1.0, 2.5] or [6.5, 8.0] depending on which bottom
    # vertex std::sort puts last: pixels (1,7) (2,7) or (7,7) (8,7) — nothing of it is inside the triangle.
    got, _ = covered(oracle_port, [(2.5, 8.5), (6.5, 8.5), (4.5, 10.5)])
    assert got in ([(1, 7), (2, 7)], [(7, 7), (8, 7)])
    # straddling the top border: the rows inside are the closed triangle, nothing is added on row 7 but its own span
    tri = [(2.25, 4.5), (7.75, 4.5), (5.0, 9.5)]
    got, _ = covered(oracle_port, tri)
    assert got == inside_closed(tri)
