"""Generate tests/golden/*.npz with the REFERENCE-backed checker (oracle/_ref).

Run in the build container (needs /root/reference to have been compiled by oracle/Makefile):

    python tests/golden/make_golden.py

Every fixture stores its inputs (grid, scalars, rotation list, image size, bounds, alpha_limit)
and the outputs of the reference's own line.cpp / tetra.cpp object code driven by
oracle/ref_driver.cpp: the fp32 image [Y, X, 2], the segment count S
(plane::count_all_intersections, plane.cpp:3-12) and the covered-pixel count.  The reference
ships no fixtures of its own (no tests, data files git-ignored), so these are the golden
vectors of the path.  Fixtures are data only.

    python tests/golden/make_golden.py g7_ g8_     # only the fixtures whose name starts with one of these
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from course5_amd import meshgen as mg  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
VIEWS = ((0.0, 0.0), (0.1, 0.07), (0.5, 0.25))
# grids with hanging nodes: the interface of g7 is (about) the plane x = const of the object, edge-on to the rays at
# Y = 0: one generic view, two with the interface steep against the rays, one oblique from behind
HANGING_VIEWS = ((0.1, 0.07), (0.3, 0.02), (0.0, 0.004), (0.37, -0.61))


def special_alpha(n):
    """Exercises every branch of line.cpp:213-224: 0, < DBL_EPSILON, tiny, > alpha_limit, huge."""
    vals = np.array([0.0, 1e-17, 2.0e-16, 2.3e-16, 1e-9, 0.5, 2.5, 2.5000001, 3.9, 1e3])
    return vals[np.arange(n) % len(vals)]


def hanging_node_grid():
    """The G2 grid with three of its cells cut in two through the midpoint of one edge each (a different edge of
    the cell every time): three single hanging nodes; every other cell round those edges keeps its faces."""
    xyz, cells = mg.kuhn_box(4, jitter=0.1)
    for cell, edge in ((100, (0, 1)), (211, (1, 3)), (37, (0, 3))):
        xyz, cells = mg.split_cell_at_edge_midpoint(xyz, cells, cell, edge)
    return xyz, cells


def main():
    only = tuple(sys.argv[1:])
    ref = Oracle("reference")
    fixtures = []
    # G1: 8-tet cube, full 60x45 images + subsampled 600x450
    xyz, cells, a, q = mg.workload("c1")
    fixtures.append(("g1_cube8_60x45", xyz, cells, a, q, 60, 45, 2.5, 1))
    fixtures.append(("g1_cube8_600x450", xyz, cells, a, q, 600, 450, 2.5, 10))
    # G2: 4^3 Kuhn grid with jitter
    xyz, cells, a, q = mg.workload("g2")
    fixtures.append(("g2_kuhn4_120x90", xyz, cells, a, q, 120, 90, 2.5, 1))
    fixtures.append(("g2_kuhn4_limit3_120x90", xyz, cells, a, q, 120, 90, 3.0, 1))
    # G4: emission/absorption branches (alpha = 0, < eps, > limit ...)
    fixtures.append(("g4_kuhn4_special_alpha_120x90", xyz, cells, special_alpha(len(cells)), q, 120, 90, 2.5, 1))
    # non-convex staircase ball (ray re-entry), small
    xyz, cells = mg.ball(12, 0.45)
    a, q = mg.scalars(len(cells))
    fixtures.append(("ball12_150x112", xyz, cells, a, q, 150, 112, 2.5, 1))

    # G7 / G8 (round 3): conforming in space, not in connectivity — the reference never looks at connectivity
    # (object3d_base.cpp:37-42 copies four points per cell, plane.cpp:184-192 bins every face, line.cpp:138 sorts)
    xyz, cells, _ = mg.refined_interface(3, 2, 3, jitter=0.1, warp=0.08)
    a, q = mg.scalars(len(cells), seed=3)
    fixtures.append(("g7_refined_interface_160x120", xyz, cells, a, q, 160, 120, 2.5, 1, HANGING_VIEWS))
    xyz, cells = hanging_node_grid()
    a, q = mg.scalars(len(cells), seed=4)
    fixtures.append(("g8_hanging_nodes_120x90", xyz, cells, a, q, 120, 90, 2.5, 1, HANGING_VIEWS))

    for fx in fixtures:
        name, xyz, cells, a, q, rx, ry, limit, stride = fx[:9]
        views = fx[9] if len(fx) > 9 else VIEWS
        if only and not name.startswith(only):
            continue
        out = dict(xyz=xyz, cells=cells.astype(np.int32), alpha=a, q=q, res=np.array([rx, ry]),
                   bounds=np.array(mg.REFERENCE_BOUNDS), alpha_limit=np.array(limit),
                   views=np.array(views), stride=np.array(stride))
        for k, (ax, ay) in enumerate(views):
            rots = mg.view_rotations(ax, ay)
            r = ref.render(xyz, cells, a, q, rots, rx, ry, mg.REFERENCE_BOUNDS, alpha_limit=limit)
            img = r["image"]
            out[f"rots{k}"] = rots
            out[f"image{k}"] = img[::stride, ::stride].copy()
            out[f"sha256_{k}"] = np.frombuffer(hashlib.sha256(img.tobytes()).digest(), dtype=np.uint8)
            out[f"segments{k}"] = np.array(r["segments"])
            out[f"covered{k}"] = np.array(r["covered"])
            print(name, (ax, ay), "S", r["segments"], "covered", r["covered"], "max", img.max(axis=(0, 1)))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)

    if only:
        return
    # G3: per-pixel segment lists of the reference's line::calculate_intersections (line.cpp:84-148) for 24 pixels of
    # the G2 fixture, view (0.1, 0.07): [(tetra id, delta z)] in the order std::sort leaves them
    xyz, cells, a, q = mg.workload("g2")
    rots = mg.view_rotations(0.1, 0.07)
    full = ref.render(xyz, cells, a, q, rots, 120, 90, mg.REFERENCE_BOUNDS)["image"]
    cov = np.argwhere(full[..., 0] > 0)  # (row, col)
    pick = cov[np.linspace(0, len(cov) - 1, 24).astype(int)]
    probes = np.stack([pick[:, 1], pick[:, 0]], axis=1).astype(np.int32)
    lists = ref.probe_segments(xyz, cells, a, q, rots, 120, 90, mg.REFERENCE_BOUNDS, probes)
    g3 = dict(probes=probes, rots=rots, res=np.array([120, 90]), bounds=np.array(mg.REFERENCE_BOUNDS),
              counts=np.array([len(l) for l in lists]))
    for k, l in enumerate(lists):
        g3[f"tet{k}"] = l[:, 0].astype(np.int64)
        g3[f"dz{k}"] = l[:, 1]
    np.savez_compressed(os.path.join(HERE, "g3_segments_g2_view1.npz"), **g3)
    print("G3:", len(lists), "pixels,", int(g3["counts"].sum()), "segments")

    # rotation known-answer vectors (tetra.cpp:44-62 via the reference's tetra::rotate_around_*)
    rng = np.random.default_rng(7)
    pts = rng.uniform(-2, 2, (64, 3))
    rot_out = {}
    for k, (ax, ay) in enumerate(VIEWS + ((1.3, -0.77),)):
        rots = mg.view_rotations(ax, ay, 0.25 * k)
        rot_out[f"rots{k}"] = rots
        rot_out[f"out{k}"] = ref.rotate_points(pts, rots)
    np.savez_compressed(os.path.join(HERE, "rotations.npz"), pts=pts, **rot_out)


if __name__ == "__main__":
    main()
