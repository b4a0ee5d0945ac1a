"""The C-ABI shared library loads and exports exactly what include/course5_hip.h declares.
No compute calls here (CPU box has no GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from course5_amd import capi, meshgen as mg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "course5_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(c5_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(capi.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.c5_abi_version() == 2


def test_status_codes_match_header():
    text = open(os.path.join(ROOT, "include", "course5_hip.h")).read()
    for name in ("C5_OK", "C5_ERR_INVALID", "C5_ERR_STATE", "C5_ERR_HIP", "C5_ERR_MESH", "C5_ERR_NO_DEVICE",
                 "C5_ERR_WALK", "C5_RETRY"):
        m = re.search(name + r"\s*=\s*(\d+)", text)
        assert m and int(m.group(1)) == getattr(capi, name)


def test_struct_layouts():
    assert ctypes.sizeof(capi.Rotation) == 24
    assert ctypes.sizeof(capi.Stats) == 6 * 8 + 2 * 4 + 6 * 4 + 8 + 2 * 8


def test_face_adjacency_helper_runs_without_a_gpu():
    xyz, cells = mg.kuhn_box(3)
    adj, n_boundary = capi.face_adjacency(cells, len(xyz))
    assert n_boundary == 6 * 3 * 3 * 2          # two triangles per boundary square
    assert (adj >= -1).all() and (adj < len(cells)).all()
    # symmetry: if b is a's neighbour across some face, a is b's neighbour across some face
    for a in range(len(cells)):
        for b in adj[a]:
            if b >= 0:
                assert a in adj[b]
    # each interior face joins cells sharing exactly three points
    a, b = 0, adj[0][adj[0] >= 0][0]
    assert len(set(cells[a]) & set(cells[b])) == 3


def test_non_conforming_grid_is_reported_by_the_adjacency_helper():
    xyz, cells = mg.cube8()
    dup = np.vstack([cells, cells[:1], cells[:1]])  # the same cell three times -> faces shared by 3 cells
    with pytest.raises(capi.C5Error) as e:
        capi.face_adjacency(dup, len(xyz))
    assert e.value.code == capi.C5_ERR_MESH


def test_point_welding_restores_the_adjacency_of_per_cell_point_copies():
    """The reference copies four points per cell and never looks at ids (object3d_base.cpp:37-42): a file
    whose cells carry private copies of their points renders there like any other.  c5_upload_grid welds
    coincident points first (c5_weld_points), so such a soup gets the same adjacency as the indexed grid."""
    xyz, cells = mg.kuhn_box(5, jitter=0.1)
    soup_xyz, soup_cells = mg.per_cell_point_copies(xyz, cells)
    assert len(soup_xyz) == 4 * len(cells)
    rep, merged = capi.weld_points(soup_xyz)
    assert merged == len(soup_xyz) - len(xyz)
    assert (rep <= np.arange(len(rep))).all() and np.array_equal(soup_xyz[rep], soup_xyz)
    # unwelded: every face is a boundary face; welded: the indexed grid's table
    _, nb_raw = capi.face_adjacency(soup_cells, len(soup_xyz))
    assert nb_raw == 4 * len(cells)
    adj_w, nb_w = capi.face_adjacency(rep[soup_cells], len(soup_xyz))
    adj, nb = capi.face_adjacency(cells, len(xyz))
    assert nb_w == nb and np.array_equal(adj_w, adj)
    # nothing coincides in an indexed grid: the identity; -0.0 and 0.0 are one coordinate
    rep0, merged0 = capi.weld_points(xyz)
    assert merged0 == 0 and np.array_equal(rep0, np.arange(len(xyz)))
    rep1, merged1 = capi.weld_points(np.array([[0.0, 1.0, 2.0], [3.0, 1.0, 2.0], [-0.0, 1.0, 2.0]]))
    assert merged1 == 1 and rep1.tolist() == [0, 1, 0]


def test_a_cell_naming_a_point_twice_is_not_walkable():
    xyz, cells = mg.cube8()
    bad = cells.copy()
    bad[3, 1] = bad[3, 0]
    with pytest.raises(capi.C5Error) as e:
        capi.face_adjacency(bad, len(xyz))
    assert e.value.code == capi.C5_ERR_MESH and "twice" in e.value.message


def test_create_without_gpu_fails_loudly():
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.C5Error) as e:
        capi.Context(0)
    assert e.value.code == capi.C5_ERR_NO_DEVICE


def test_face_adjacency_of_a_large_grid_matches_numpy():
    """Large enough for the parallel bucket sort of the face keys (adjacency.cpp: sort_keys)."""
    xyz, cells = mg.kuhn_box(28, jitter=0.1)   # 131 712 cells, 526 848 face keys
    adj, n_boundary = capi.face_adjacency(cells, len(xyz))
    fv = np.array([[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]])
    faces = np.sort(cells[:, fv].reshape(-1, 3), axis=1)            # [4n, 3], face f of cell c at 4c + f
    order = np.lexsort((faces[:, 2], faces[:, 1], faces[:, 0]))
    sf = faces[order]
    same_next = np.r_[(sf[1:] == sf[:-1]).all(axis=1), False]
    same_prev = np.r_[False, same_next[:-1]]
    want = np.full(4 * len(cells), -1, dtype=np.int64)
    i = np.nonzero(same_next)[0]
    want[order[i]] = order[i + 1] // 4
    want[order[i + 1]] = order[i] // 4
    assert int((~same_next & ~same_prev).sum()) == n_boundary == 6 * 28 * 28 * 2
    assert np.array_equal(adj.reshape(-1), want)
