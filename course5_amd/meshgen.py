"""Deterministic synthetic tetrahedral grids and legacy-VTK writers (host tooling).

The reference ships no data files (`/root/reference/.gitignore:3-4` excludes *.vtk / *.vti), so
every grid used by tests and by `bench.py` is generated here from fixed seeds.  The on-disk
contract is the one the reference reads (`project/src/object3d_base.cpp:13-53`,
`object3d_accretion_disk.cpp:4`): legacy VTK `DATASET UNSTRUCTURED_GRID`, all cells
tetrahedra, CELL_DATA scalars named `AbsorpCoef` and `radEnLooseRate`.

Workloads follow SURVEY.md §8(d):
  C1  cube8()                 8 tets, 9 points, inside the hard-coded domain
  C2  ball(n=36, r=0.45)      ~107k tets, non-convex staircase boundary
  C3  kuhn_box(n=55)          998 250 tets, 175 616 points
"""
from __future__ import annotations

import itertools

import numpy as np

# Domain of the reference, in its own order {x_max, x_min, y_max, y_min} (main.cpp:83).
REFERENCE_BOUNDS = (2.2, -0.2, 0.9, -0.9)
# View used for every benchmark configuration (generic: avoids pixel/vertex alignment).
BENCH_VIEW = dict(angle_around_x=0.1, angle_around_y=0.07, initial_system_angle=0.0)


def signed_volumes(xyz: np.ndarray, cells: np.ndarray) -> np.ndarray:
    p = xyz[cells]  # [n,4,3]
    a = p[:, 1] - p[:, 0]
    b = p[:, 2] - p[:, 0]
    c = p[:, 3] - p[:, 0]
    return np.einsum("ij,ij->i", a, np.cross(b, c)) / 6.0


def orient_positive(xyz: np.ndarray, cells: np.ndarray) -> np.ndarray:
    """Swap the last two vertices of negatively oriented cells (in place on a copy)."""
    cells = cells.copy()
    neg = signed_volumes(xyz, cells) < 0
    cells[neg, 2], cells[neg, 3] = cells[neg, 3].copy(), cells[neg, 2].copy()
    return cells


def validate(xyz: np.ndarray, cells: np.ndarray, min_rel_volume: float = 1e-6) -> None:
    """Reject meshes a face-adjacency walk (or the reference's pairing) cannot handle."""
    vol = signed_volumes(xyz, cells)
    if not np.all(np.isfinite(vol)):
        raise ValueError("non-finite cell volume")
    scale = np.abs(vol).mean()
    if np.any(vol <= min_rel_volume * scale):
        raise ValueError(f"{int((vol <= min_rel_volume * scale).sum())} inverted or degenerate cells")


def scalars(n_cells: int, seed: int = 1234) -> tuple[np.ndarray, np.ndarray]:
    """alpha ~ U[0,4) (exercises the 2.5 clamp), Q ~ U[0,1)  (SURVEY §8(d))."""
    rng = np.random.default_rng(seed)
    alpha = rng.uniform(0.0, 4.0, n_cells)
    q = rng.uniform(0.0, 1.0, n_cells)
    return alpha, q


def cube8(lo=(0.5, -0.5, -0.5), size: float = 1.0):
    """Five-tet split of a cube whose central tet is split at its centroid: 8 tets, 9 points."""
    corners = np.array(list(itertools.product((0.0, 1.0), repeat=3)))  # id = 4x+2y+z
    vid = lambda x, y, z: 4 * x + 2 * y + z  # noqa: E731
    even = [vid(0, 0, 0), vid(1, 1, 0), vid(1, 0, 1), vid(0, 1, 1)]
    corner_tets = [
        [vid(1, 0, 0), vid(0, 0, 0), vid(1, 1, 0), vid(1, 0, 1)],
        [vid(0, 1, 0), vid(0, 0, 0), vid(1, 1, 0), vid(0, 1, 1)],
        [vid(0, 0, 1), vid(0, 0, 0), vid(1, 0, 1), vid(0, 1, 1)],
        [vid(1, 1, 1), vid(1, 1, 0), vid(1, 0, 1), vid(0, 1, 1)],
    ]
    centre = corners[even].mean(axis=0)
    pts = np.vstack([corners, centre])
    c = 8
    central = [[c, even[0], even[1], even[2]], [c, even[0], even[1], even[3]],
               [c, even[0], even[2], even[3]], [c, even[1], even[2], even[3]]]
    cells = np.array(corner_tets + central, dtype=np.int32)
    xyz = np.asarray(lo, dtype=np.float64) + size * pts
    cells = orient_positive(xyz, cells)
    validate(xyz, cells)
    return xyz, cells


_KUHN_PERMS = list(itertools.permutations(range(3)))


def kuhn_box(n: int, lo=(0.5, -0.5, -0.5), size: float = 1.0, jitter: float = 0.0, seed: int = 1234,
             keep=None):
    """n^3 cubes x 6 Kuhn tets (conforming).  Interior points get uniform jitter <= jitter*h.

    keep: optional callable(centroids[n_cells,3]) -> bool mask selecting cells to keep
    (points are compacted afterwards).
    """
    h = size / n
    m = n + 1
    g = np.arange(m)
    ii, jj, kk = np.meshgrid(g, g, g, indexing="ij")
    xyz = np.stack([ii, jj, kk], axis=-1).reshape(-1, 3).astype(np.float64) * h
    if jitter > 0:
        rng = np.random.default_rng(seed)
        d = rng.uniform(-jitter * h, jitter * h, xyz.shape)
        interior = ((ii > 0) & (ii < n) & (jj > 0) & (jj < n) & (kk > 0) & (kk < n)).reshape(-1)
        xyz[interior] += d[interior]
    xyz += np.asarray(lo, dtype=np.float64)

    ci, cj, ck = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    base = np.stack([ci, cj, ck], axis=-1).reshape(-1, 3)  # [n^3,3]
    pid = lambda c: (c[:, 0] * m + c[:, 1]) * m + c[:, 2]  # noqa: E731
    cells = []
    for perm in _KUHN_PERMS:
        v = [base.copy()]
        for axis in perm:
            nxt = v[-1].copy()
            nxt[:, axis] += 1
            v.append(nxt)
        cells.append(np.stack([pid(c) for c in v], axis=1))
    cells = np.stack(cells, axis=1).reshape(-1, 4).astype(np.int32)  # cube-major, 6 per cube

    if keep is not None:
        mask = keep(xyz[cells].mean(axis=1))
        cells = cells[mask]
        used = np.unique(cells)
        remap = np.full(xyz.shape[0], -1, dtype=np.int64)
        remap[used] = np.arange(used.size)
        xyz = xyz[used]
        cells = remap[cells].astype(np.int32)

    cells = orient_positive(xyz, cells)
    validate(xyz, cells)
    return xyz, cells


def ball(n: int = 36, r: float = 0.45, centre=(1.0, 0.0, 0.0), jitter: float = 0.1, seed: int = 1234):
    """Kuhn box cells whose centroid lies within r of centre: non-convex staircase boundary (C2)."""
    c = np.asarray(centre, dtype=np.float64)
    return kuhn_box(n, jitter=jitter, seed=seed,
                    keep=lambda cen: np.linalg.norm(cen - c, axis=1) < r)


def _kuhn_lattice(nx: int, ny: int, nz: int):
    """Integer lattice points [(nx+1)(ny+1)(nz+1), 3] and the 6 Kuhn tets of every cube (ids into them)."""
    gi, gj, gk = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), np.arange(nz + 1), indexing="ij")
    pts = np.stack([gi, gj, gk], axis=-1).reshape(-1, 3)
    ci, cj, ck = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    base = np.stack([ci, cj, ck], axis=-1).reshape(-1, 3)
    pid = lambda c: (c[:, 0] * (ny + 1) + c[:, 1]) * (nz + 1) + c[:, 2]  # noqa: E731
    cells = []
    for perm in _KUHN_PERMS:
        v = [base.copy()]
        for axis in perm:
            nxt = v[-1].copy()
            nxt[:, axis] += 1
            v.append(nxt)
        cells.append(np.stack([pid(c) for c in v], axis=1))
    return pts, np.stack(cells, axis=1).reshape(-1, 4).astype(np.int32)


def refined_interface(n: int = 3, nx_coarse: int = 2, nx_fine: int = 3, lo=(0.55, -0.4, -0.4), size: float = 0.8,
                      jitter: float = 0.1, warp: float = 0.0, seed: int = 1234, weld: bool = True):
    """A coarse Kuhn box (nx_coarse x n x n cubes of edge h = size / n) abutting a 2x-refined one
    (nx_fine x 2n x 2n cubes of edge h / 2) across the plane x = lo.x + nx_coarse h: a grid that is conforming in
    SPACE but not in connectivity.  Every coarse interface triangle is the union of four coplanar fine ones
    (the fine interface nodes are the coarse nodes, the midpoints of the coarse edges and of the coarse
    diagonals: hanging nodes), so no face of the interface has a partner with the same three points.  The
    reference renders such grids like any other — it copies four points per cell and never looks at
    connectivity (object3d_base.cpp:37-42), bins every face (plane.cpp:184-192) and sorts (line.cpp:138).

    jitter: uniform displacement (units of the own cube edge) of the nodes strictly inside either box.
    warp:   displacement (units of h, all three directions) of the coarse interface nodes off the rim, so that
            the interface is a crumpled surface and not a plane; the hanging nodes follow (midpoints).
    weld:   True -> coincident interface nodes share one id; False -> the two boxes keep their own points
            (the library welds by coordinate, c5_weld_points).
    Returns (xyz, cells, n_coarse_cells): cells [0, n_coarse_cells) are the coarse box.
    """
    rng = np.random.default_rng(seed)
    h = size / n
    lo = np.asarray(lo, dtype=np.float64)
    # coarse box
    pa, ca = _kuhn_lattice(nx_coarse, n, n)
    xa = pa.astype(np.float64) * h
    inner_a = (pa[:, 0] > 0) & (pa[:, 0] < nx_coarse) & (pa[:, 1] > 0) & (pa[:, 1] < n) & (pa[:, 2] > 0) & (pa[:, 2] < n)
    xa[inner_a] += rng.uniform(-jitter * h, jitter * h, (int(inner_a.sum()), 3))
    face_a = (pa[:, 0] == nx_coarse) & (pa[:, 1] > 0) & (pa[:, 1] < n) & (pa[:, 2] > 0) & (pa[:, 2] < n)
    if warp > 0:
        xa[face_a] += rng.uniform(-warp * h, warp * h, (int(face_a.sum()), 3))
    xa += lo
    # fine box
    pb, cb = _kuhn_lattice(nx_fine, 2 * n, 2 * n)
    hf = h / 2
    xb = pb.astype(np.float64) * hf
    xb[:, 0] += nx_coarse * h
    inner_b = (pb[:, 0] > 0) & (pb[:, 0] < nx_fine) & (pb[:, 1] > 0) & (pb[:, 1] < 2 * n) & (pb[:, 2] > 0) & (pb[:, 2] < 2 * n)
    xb[inner_b] += rng.uniform(-jitter * hf, jitter * hf, (int(inner_b.sum()), 3))
    xb += lo
    # its interface nodes: coarse nodes, edge midpoints, diagonal midpoints (the Kuhn diagonal of a y-z square runs
    # from (j, k) to (j + 1, k + 1) on every x = const face)
    coarse_id = lambda j, k: (nx_coarse * (n + 1) + j) * (n + 1) + k  # noqa: E731
    on_face = np.nonzero(pb[:, 0] == 0)[0]
    J, K = pb[on_face, 1], pb[on_face, 2]
    j0, j1 = J // 2, (J + 1) // 2
    k0, k1 = K // 2, (K + 1) // 2
    xb[on_face] = 0.5 * (xa[coarse_id(j0, k0)] + xa[coarse_id(j1, k1)])  # (x + x) / 2 == x for the even-even nodes
    xyz = np.vstack([xa, xb])
    cells = np.vstack([ca, cb + len(xa)]).astype(np.int32)
    if weld:
        same = on_face[(J % 2 == 0) & (K % 2 == 0)]
        remap = np.arange(len(xyz))
        remap[len(xa) + same] = coarse_id(pb[same, 1] // 2, pb[same, 2] // 2)
        cells = remap[cells]
        used = np.unique(cells)
        new_id = np.full(len(xyz), -1, dtype=np.int64)
        new_id[used] = np.arange(used.size)
        xyz = xyz[used]
        cells = new_id[cells].astype(np.int32)
    cells = orient_positive(xyz, cells)
    validate(xyz, cells)
    return xyz, cells, len(ca)


def split_cell_at_edge_midpoint(xyz: np.ndarray, cells: np.ndarray, cell: int, edge=(0, 1)):
    """Cut ONE cell in two through the midpoint of one of its edges and leave every other cell round that edge
    as it is: a single hanging node.  The two faces of the cell that contain the edge become two faces each,
    and the neighbours behind them keep their one face — conforming in space, not in connectivity.
    Returns (xyz', cells') with the two halves in place of / behind the cut cell (ids of other cells unchanged)."""
    xyz = np.asarray(xyz, dtype=np.float64)
    cells = np.asarray(cells, dtype=np.int32).copy()
    a, b = int(cells[cell][edge[0]]), int(cells[cell][edge[1]])
    m = len(xyz)
    xyz = np.vstack([xyz, 0.5 * (xyz[a] + xyz[b])])
    first, second = cells[cell].copy(), cells[cell].copy()
    first[edge[1]] = m   # (a, m, ., .)
    second[edge[0]] = m  # (m, b, ., .)
    cells[cell] = first
    cells = np.vstack([cells, second[None, :]]).astype(np.int32)
    cells = orient_positive(xyz, cells)
    validate(xyz, cells)
    return xyz, cells


def per_cell_point_copies(xyz: np.ndarray, cells: np.ndarray):
    """The same grid as a soup: every cell gets four private points (what object3d_base::read_vtk_file
    keeps of a file, object3d_base.cpp:37-42, and what some writers emit).  Returns (xyz', cells')."""
    cells = np.asarray(cells)
    soup_xyz = np.ascontiguousarray(np.asarray(xyz, dtype=np.float64)[cells.reshape(-1)])
    soup_cells = np.arange(4 * len(cells), dtype=np.int32).reshape(-1, 4)
    return soup_xyz, soup_cells


def workload(name: str):
    """Named benchmark / test grids -> (xyz, cells, alpha, q)."""
    if name == "c1":
        xyz, cells = cube8()
    elif name == "g2":  # 4^3 Kuhn grid, 384 tets
        xyz, cells = kuhn_box(4, jitter=0.1)
    elif name == "c2":
        xyz, cells = ball(36, 0.45)
    elif name == "c3":
        xyz, cells = kuhn_box(55, jitter=0.1)
    elif name == "refined":  # coarse box against a 2x-refined one: hanging nodes all over the interface
        xyz, cells, _ = refined_interface(3, 2, 3, jitter=0.1, warp=0.08)
    elif name.startswith("kuhn"):
        xyz, cells = kuhn_box(int(name[4:]), jitter=0.1)
    else:
        raise ValueError(f"unknown workload {name!r}")
    alpha, q = scalars(cells.shape[0])
    return xyz, cells, alpha, q


def view_rotations(angle_around_x: float = 0.0, angle_around_y: float = 0.0,
                   initial_system_angle: float = 0.0, x0: float = 1.0) -> np.ndarray:
    """Rotation list [n,3] = {axis (0 = x, 1 = y), angle, x0} applied to the grid by `course`.

    Mirrors `/root/reference/project/src/main.cpp:96,105-107`; angles are in units of pi.
    """
    pi = 3.14159265358979323846  # config.hpp:45
    mp = -initial_system_angle * pi + pi / 2.0
    return np.array([[0.0, mp, 0.0],
                     [1.0, angle_around_y * pi, x0],
                     [0.0, -mp + angle_around_x * pi, 0.0]], dtype=np.float64)


def write_vtk_ascii(path: str, xyz: np.ndarray, cells: np.ndarray, alpha: np.ndarray, q: np.ndarray) -> None:
    """Legacy ASCII VTK unstructured grid with the two cell scalars the reference reads."""
    n_pts, n_cells = xyz.shape[0], cells.shape[0]
    with open(path, "w") as f:
        f.write("# vtk DataFile Version 3.0\nsynthetic tetrahedral grid\nASCII\nDATASET UNSTRUCTURED_GRID\n")
        f.write(f"POINTS {n_pts} double\n")
        np.savetxt(f, xyz, fmt="%.17g")
        f.write(f"CELLS {n_cells} {5 * n_cells}\n")
        np.savetxt(f, np.hstack([np.full((n_cells, 1), 4, dtype=np.int64), cells.astype(np.int64)]), fmt="%d")
        f.write(f"CELL_TYPES {n_cells}\n")
        np.savetxt(f, np.full(n_cells, 10, dtype=np.int64), fmt="%d")
        f.write(f"CELL_DATA {n_cells}\n")
        for name, arr in (("AbsorpCoef", alpha), ("radEnLooseRate", q)):
            f.write(f"SCALARS {name} double 1\nLOOKUP_TABLE default\n")
            np.savetxt(f, arr, fmt="%.17g")


def write_vtk_binary(path: str, xyz, cells, alpha, q, v51: bool = False) -> None:
    """Legacy BINARY (big-endian) variant; v51=True uses the 5.1 OFFSETS/CONNECTIVITY cell layout."""
    n_pts, n_cells = xyz.shape[0], cells.shape[0]
    with open(path, "wb") as f:
        f.write(("# vtk DataFile Version %s\nsynthetic tetrahedral grid\nBINARY\nDATASET UNSTRUCTURED_GRID\n"
                 % ("5.1" if v51 else "3.0")).encode())
        f.write(f"POINTS {n_pts} double\n".encode())
        f.write(np.ascontiguousarray(xyz, dtype=">f8").tobytes())
        f.write(b"\n")
        if v51:
            f.write(f"CELLS {n_cells + 1} {4 * n_cells}\nOFFSETS vtktypeint64\n".encode())
            f.write((4 * np.arange(n_cells + 1)).astype(">i8").tobytes())
            f.write(b"\nCONNECTIVITY vtktypeint64\n")
            f.write(np.ascontiguousarray(cells).astype(">i8").tobytes())
        else:
            f.write(f"CELLS {n_cells} {5 * n_cells}\n".encode())
            f.write(np.hstack([np.full((n_cells, 1), 4), cells]).astype(">i4").tobytes())
        f.write(f"\nCELL_TYPES {n_cells}\n".encode())
        f.write(np.full(n_cells, 10).astype(">i4").tobytes())
        f.write(f"\nCELL_DATA {n_cells}\n".encode())
        # the second array goes through FIELD data, which newer writers use for extra arrays
        f.write(b"SCALARS AbsorpCoef double 1\nLOOKUP_TABLE default\n")
        f.write(np.ascontiguousarray(alpha, dtype=">f8").tobytes())
        f.write(f"\nFIELD FieldData 1\nradEnLooseRate 1 {n_cells} float\n".encode())
        f.write(np.ascontiguousarray(q).astype(">f4").tobytes())
        f.write(b"\n")
