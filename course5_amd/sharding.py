"""Row-tile sharding of the image plane across ranks (host logic, mirrors device_types.hpp).

Rows are grouped in tiles of `tile_rows`; tile t belongs to rank t % world.  A rank's local strip
holds its rows in ascending global order.  Pixels are independent in the reference
(plane.cpp:161-169 has no cross-pixel state), so the only exchange is one gather of strips at
frame end.
"""
from __future__ import annotations

import numpy as np


def local_rows(res_y: int, tile_rows: int, rank: int, world: int) -> np.ndarray:
    """Global row indices owned by `rank`, ascending."""
    rows = np.arange(res_y)
    return rows[(rows // tile_rows) % world == rank]


def local_row_count(res_y: int, tile_rows: int, rank: int, world: int) -> int:
    return int(local_rows(res_y, tile_rows, rank, world).size)


def padded_rows(res_y: int, tile_rows: int, world: int) -> int:
    """Strip height every rank pads to so that an all_gather / gather has equal-sized parts."""
    return max(local_row_count(res_y, tile_rows, r, world) for r in range(world))


def assemble(strips, res_y: int, tile_rows: int, world: int) -> np.ndarray:
    """strips[r]: [>= local_row_count(r), res_x, 2] -> full image [res_y, res_x, 2]."""
    res_x = strips[0].shape[1]
    out = np.empty((res_y, res_x, 2), dtype=strips[0].dtype)
    for r in range(world):
        rows = local_rows(res_y, tile_rows, r, world)
        out[rows] = np.asarray(strips[r])[: rows.size]
    return out
