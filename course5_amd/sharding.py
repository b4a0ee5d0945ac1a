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


# ---------------------------------------------------------------------------------------------
# contiguous, cost-balanced row blocks (one block per rank)
# ---------------------------------------------------------------------------------------------
def equal_blocks(res_y: int, world: int):
    """[(begin, count)] per rank: res_y rows split as evenly as possible."""
    edges = [(r * res_y) // world for r in range(world + 1)]
    return [(edges[r], edges[r + 1] - edges[r]) for r in range(world)]


def balanced_blocks(row_costs, world: int, base_cost: float = 0.0, min_rows: int = 1, quantum: int = 1):
    """Split rows into `world` contiguous blocks of (nearly) equal total cost.

    row_costs[r] = measured cost of row r (segments of the previous frame); base_cost is added per
    row for the work every pixel costs regardless (entry lookup, store).  Every block gets at least
    min_rows rows.  Deterministic, so every rank computes the same partition from the same costs.
    quantum > 1: cuts at multiples of `quantum` rows (the walk's tiles are 8 rows tall and counted from a block's first
    row: a block of 241 rows costs 31 rows of tiles, the last of them one pixel tall - an eighth of the 4800x3600 frame
    0.319 -> 0.337 ms); ignored for images too small to give every rank two quanta.
    """
    c = np.asarray(row_costs, dtype=np.float64) + float(base_cost)
    res_y = c.size
    if res_y < world * min_rows:
        raise ValueError("more ranks than rows")
    cum = np.concatenate([[0.0], np.cumsum(c)])
    total = cum[-1]
    edges = [0]
    for r in range(1, world):
        target = total * r / world
        e = int(np.searchsorted(cum, target, side="left"))
        # nearest of the two candidate cut positions
        if e > 0 and abs(cum[e - 1] - target) <= abs(cum[min(e, res_y)] - target):
            e -= 1
        if quantum > 1 and res_y >= 2 * quantum * world:
            e = int(e + quantum // 2) // quantum * quantum
        e = max(e, edges[-1] + min_rows)
        e = min(e, res_y - (world - r) * min_rows)
        edges.append(e)
    edges.append(res_y)
    return [(edges[r], edges[r + 1] - edges[r]) for r in range(world)]


def time_weighted_costs(row_costs, blocks, times, base_cost: float = 0.0) -> np.ndarray:
    """Row costs rescaled block by block so that every block's total is the TIME its rank took for it.

    The model (segments + base_cost per row) misses what a row's rays cost per segment - the tiles of an oblique face step
    half as fast as the tiles of the grid's body - and what a share costs whatever its rows: with the times of one frame on
    the model's blocks at hand, balanced_blocks(time_weighted_costs(...), world) cuts the rows by measured time instead
    (the same arithmetic as csrc/host/row_blocks.cpp: time_weighted_row_costs)."""
    c = np.asarray(row_costs, dtype=np.float64) + float(base_cost)
    out = c.copy()
    for (b, n), t in zip(blocks, times):
        s = float(c[b:b + n].sum())
        if n > 0 and s > 0.0 and t > 0.0:
            out[b:b + n] = c[b:b + n] * (float(t) / s)
    return out


def assemble_blocks(strips, blocks, res_y: int) -> np.ndarray:
    res_x = strips[0].shape[1]
    out = np.empty((res_y, res_x, 2), dtype=strips[0].dtype)
    for s, (b, n) in zip(strips, blocks):
        out[b:b + n] = np.asarray(s)[:n]
    return out
