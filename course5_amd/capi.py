"""ctypes binding of the C ABI in include/course5_hip.h (libcourse5_hip.so).

Test/bench plumbing only: the product is the shared library and the `course` CLI.  There is no
CPU fallback here — if the library is missing or no GPU is present, calls raise.

When device buffers are shared with PyTorch (render_device into a torch tensor), import torch
BEFORE the first call into this module: torch bundles its own HIP runtime, and whichever runtime is
loaded first serves the whole process.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("C5_LIB", os.path.join(PKG, "libcourse5_hip.so"))  # C5_LIB: A/B another build

C5_OK, C5_ERR_INVALID, C5_ERR_STATE, C5_ERR_HIP, C5_ERR_MESH, C5_ERR_NO_DEVICE, C5_ERR_WALK, C5_RETRY = range(8)

# every symbol include/course5_hip.h declares
EXPORTS = [
    "c5_abi_version", "c5_device_count", "c5_create", "c5_destroy", "c5_last_error",
    "c5_upload_grid", "c5_update_scalars", "c5_set_solid", "c5_set_image", "c5_set_row_tiles",
    "c5_local_rows", "c5_set_view", "c5_set_solid_view", "c5_set_alpha_limit", "c5_set_option",
    "c5_render", "c5_render_device", "c5_synchronize", "c5_get_stats", "c5_walk_kernel_ms",
    "c5_download_view_points", "c5_face_adjacency", "c5_set_stream",
    "c5_set_row_range", "c5_get_row_costs", "c5_weld_points",
    "c5_render_host_async", "c5_render_host_wait", "c5_host_alloc", "c5_host_free", "c5_render_frame_rows_async",
]


class Rotation(C.Structure):
    _fields_ = [("axis", C.c_int32), ("reserved", C.c_int32), ("angle", C.c_double), ("x0", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("segments", C.c_int64), ("covered_pixels", C.c_int64), ("solid_pixels", C.c_int64),
                ("entries", C.c_int64), ("boundary_faces", C.c_int64), ("steps", C.c_int64),
                ("walk_overflow", C.c_int32), ("entry_overflow", C.c_int32),
                ("ms_transform", C.c_float), ("ms_records", C.c_float), ("ms_entries", C.c_float),
                ("ms_solids", C.c_float), ("ms_walk", C.c_float), ("ms_total", C.c_float),
                ("odd_pixels", C.c_int64), ("pool_entries", C.c_int64), ("pool_capacity", C.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class C5Error(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"[c5 status {code}] {message}")
        self.code = code
        self.message = message


_lib = None


def load_library() -> C.CDLL:
    """dlopen the in-tree library (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} is missing: run `python -m course5_amd.build` (or __graft_entry__.build())")
    lib = C.CDLL(LIB_PATH)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p
    lib.c5_abi_version.restype = C.c_int
    lib.c5_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.c5_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.c5_destroy.argtypes = [vp]
    lib.c5_destroy.restype = None
    lib.c5_last_error.argtypes = [vp]
    lib.c5_last_error.restype = C.c_char_p
    lib.c5_upload_grid.argtypes = [vp, dp, C.c_int64, ip, C.c_int64, dp, dp]
    lib.c5_update_scalars.argtypes = [vp, dp, dp, C.c_int64]
    lib.c5_set_solid.argtypes = [vp, C.c_int, dp, C.c_int64, C.c_double]
    lib.c5_set_image.argtypes = [vp, C.c_int, C.c_int, dp]
    lib.c5_set_row_tiles.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    lib.c5_local_rows.argtypes = [vp, C.POINTER(C.c_int)]
    lib.c5_set_view.argtypes = [vp, C.POINTER(Rotation), C.c_int]
    lib.c5_set_solid_view.argtypes = [vp, C.c_int, C.POINTER(Rotation), C.c_int]
    lib.c5_set_alpha_limit.argtypes = [vp, C.c_double]
    lib.c5_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    lib.c5_render.argtypes = [vp, C.POINTER(C.c_float)]
    lib.c5_render_device.argtypes = [vp, vp]
    lib.c5_synchronize.argtypes = [vp]
    lib.c5_get_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.c5_walk_kernel_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.c5_download_view_points.argtypes = [vp, dp]
    lib.c5_set_stream.argtypes = [vp, vp]
    lib.c5_set_row_range.argtypes = [vp, C.c_int, C.c_int]
    lib.c5_get_row_costs.argtypes = [vp, C.POINTER(C.c_uint32), C.c_int]
    lib.c5_face_adjacency.argtypes = [ip, C.c_int64, C.c_int64, ip, C.POINTER(C.c_int64)]
    lib.c5_weld_points.argtypes = [dp, C.c_int64, ip, C.POINTER(C.c_int64)]
    lib.c5_render_host_async.argtypes = [vp, C.POINTER(C.c_float)]
    lib.c5_render_host_wait.argtypes = [vp]
    lib.c5_render_frame_rows_async.argtypes = [vp, C.POINTER(C.c_float)]
    lib.c5_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.c5_host_free.argtypes = [vp, vp]
    for name in EXPORTS:
        if name not in ("c5_destroy", "c5_last_error"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def _rot_array(rots) -> tuple:
    rots = np.asarray(rots, dtype=np.float64).reshape(-1, 3)
    arr = (Rotation * max(len(rots), 1))()
    for k, (axis, angle, x0) in enumerate(rots):
        arr[k].axis = int(axis)
        arr[k].angle = float(angle)
        arr[k].x0 = float(x0)
    return arr, len(rots)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Context:
    """One GPU render context (c5_context)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        self.handle = C.c_void_p()
        rc = self.lib.c5_create(device, C.byref(self.handle))
        if rc != C5_OK:
            raise C5Error(rc, self.lib.c5_last_error(None).decode())
        self.res_x = self.res_y = 0
        # the library's instruments are off by default (they cost 5 % of a frame); tests and scripts read stats()["ms_*"] and
        # walk_kernel_ms() everywhere, so this wrapper switches them on - bench.py switches the stage events off again
        self.set_option("stage_timing", 1)
        self.set_option("walk_timing", 1)

    def close(self):
        if getattr(self, "handle", None) and self.handle.value:
            self.lib.c5_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int, allow=()):
        if rc != C5_OK and rc not in allow:
            raise C5Error(rc, self.lib.c5_last_error(self.handle).decode())
        return rc

    def set_stream(self, stream_ptr: int):
        """Run on a caller-owned HIP stream (e.g. torch.cuda.current_stream().cuda_stream); 0 = own."""
        return self._check(self.lib.c5_set_stream(self.handle, C.c_void_p(stream_ptr)), allow=(C5_RETRY,))

    # -- scene ---------------------------------------------------------------------------------
    def upload_grid(self, xyz, cells, alpha, q):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        cells = np.ascontiguousarray(cells, dtype=np.int32).reshape(-1, 4)
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        q = np.ascontiguousarray(q, dtype=np.float64)
        if alpha.shape[0] != cells.shape[0] or q.shape[0] != cells.shape[0]:
            raise ValueError("one alpha and one q per cell")
        self._check(self.lib.c5_upload_grid(self.handle, _dp(xyz), xyz.shape[0],
                                            cells.ctypes.data_as(C.POINTER(C.c_int32)), cells.shape[0],
                                            _dp(alpha), _dp(q)))

    def update_scalars(self, alpha, q):
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        q = np.ascontiguousarray(q, dtype=np.float64)
        self._check(self.lib.c5_update_scalars(self.handle, _dp(alpha), _dp(q), alpha.shape[0]))

    def set_solid(self, slot: int, tets, colour: float = float("nan")):
        tets = np.ascontiguousarray(tets, dtype=np.float64).reshape(-1, 12)
        self._check(self.lib.c5_set_solid(self.handle, slot, _dp(tets), tets.shape[0], colour))

    # -- per frame -----------------------------------------------------------------------------
    def set_image(self, res_x: int, res_y: int, bounds):
        b = np.ascontiguousarray(bounds, dtype=np.float64)
        self._check(self.lib.c5_set_image(self.handle, res_x, res_y, _dp(b)))
        self.res_x, self.res_y = res_x, res_y

    def set_row_tiles(self, tile_rows: int, rank: int, world: int):
        self._check(self.lib.c5_set_row_tiles(self.handle, tile_rows, rank, world))

    def set_row_range(self, row_begin: int, row_count: int = -1):
        self._check(self.lib.c5_set_row_range(self.handle, row_begin, row_count))

    def row_costs(self) -> np.ndarray:
        """Segments per local row of the last frame (option "row_costs" must be on)."""
        out = np.zeros(self.local_rows, dtype=np.uint32)
        self._check(self.lib.c5_get_row_costs(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint32)), out.size))
        return out

    @property
    def local_rows(self) -> int:
        n = C.c_int()
        self._check(self.lib.c5_local_rows(self.handle, C.byref(n)))
        return n.value

    def set_view(self, rots):
        arr, n = _rot_array(rots)
        self._check(self.lib.c5_set_view(self.handle, arr, n))

    def set_solid_view(self, slot: int, rots):
        arr, n = _rot_array(rots)
        self._check(self.lib.c5_set_solid_view(self.handle, slot, arr, n))

    def set_alpha_limit(self, v: float):
        self._check(self.lib.c5_set_alpha_limit(self.handle, v))

    def set_option(self, name: str, value: float):
        self._check(self.lib.c5_set_option(self.handle, name.encode(), float(value)))

    # -- render --------------------------------------------------------------------------------
    def render(self) -> np.ndarray:
        """Synchronous render of the local rows -> float32 [rows, res_x, 2] on the host."""
        out = np.empty((self.local_rows, self.res_x, 2), dtype=np.float32)
        self._check(self.lib.c5_render(self.handle, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def render_device(self, device_ptr: int):
        """Asynchronous render into device memory (e.g. a torch tensor's data_ptr())."""
        self._check(self.lib.c5_render_device(self.handle, C.c_void_p(device_ptr)))

    # -- frames delivered to host memory, pipelined -------------------------------------------------
    def host_image(self, full: bool = False) -> np.ndarray:
        """A pinned float32 [local_rows (or res_y), res_x, 2] image (c5_host_alloc); release with free_host_image."""
        rows = self.res_y if full else self.local_rows
        n = rows * self.res_x * 2
        p = C.c_void_p()
        self._check(self.lib.c5_host_alloc(self.handle, n * 4, C.byref(p)))
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n,)).reshape(rows, self.res_x, 2)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def free_host_image(self, arr: np.ndarray):
        p = self._pinned.pop(arr.ctypes.data)
        self._check(self.lib.c5_host_free(self.handle, p))

    def render_host_async(self, out: np.ndarray):
        self._check(self.lib.c5_render_host_async(self.handle, out.ctypes.data_as(C.POINTER(C.c_float))))

    def render_frame_rows_async(self, frame: np.ndarray):
        """This context's rows straight into their places of the full [res_y, res_x, 2] host image."""
        self._check(self.lib.c5_render_frame_rows_async(self.handle, frame.ctypes.data_as(C.POINTER(C.c_float))))

    def render_host_wait(self) -> int:
        return self._check(self.lib.c5_render_host_wait(self.handle), allow=(C5_RETRY,))

    def bench_host_frames(self, frames: int, ring: int = 3) -> dict:
        """Throughput of frames DELIVERED TO HOST MEMORY (what plane::trace_rays returns, plane.cpp:144-172):
        c5_render_host_async / _wait with `ring` frames in flight into pinned images, plus the plain
        synchronous c5_render into a pinned and into a pageable image."""
        import time
        bufs = [self.host_image() for _ in range(ring)]
        rays = self.local_rows * self.res_x
        out = {}
        try:
            def burst(n):
                bad = False
                for k in range(n):
                    if k >= ring:
                        bad |= self.render_host_wait() != C5_OK
                    self.render_host_async(bufs[k % ring])
                for k in range(min(ring, n)):
                    bad |= self.render_host_wait() != C5_OK
                return bad

            for attempt in range(3):
                # warm: the ring's device images and pinned pages are touched, the copy stream exists, the clocks are up
                # (BENCH_r02: 1.18 ms per frame over 20 cold frames against 0.78 sustained)
                burst(max(40, ring))
                t0 = time.perf_counter()
                redo = burst(frames)
                dt = time.perf_counter() - t0
                if dt < 0.2 and not redo:  # at least 0.2 s of frames
                    frames = int(frames * 0.25 / max(dt, 1e-3)) + 1
                    t0 = time.perf_counter()
                    redo = burst(frames)
                    dt = time.perf_counter() - t0
                if not redo:
                    break
            out["pipelined"] = {"ms_per_frame": round(dt * 1e3 / frames, 4), "value": round(rays * frames / dt / 1e6, 1),
                                "frames": frames, "in_flight": ring, "destination": "pinned"}
            for name, dst in (("sync_pinned", bufs[0]), ("sync_pageable", np.empty_like(bufs[0]))):
                self._check(self.lib.c5_render(self.handle, dst.ctypes.data_as(C.POINTER(C.c_float))))
                n = max(5, frames // 5)
                t0 = time.perf_counter()
                for _ in range(n):
                    self._check(self.lib.c5_render(self.handle, dst.ctypes.data_as(C.POINTER(C.c_float))))
                dt = time.perf_counter() - t0
                out[name] = {"ms_per_frame": round(dt * 1e3 / n, 4), "value": round(rays * n / dt / 1e6, 1), "frames": n}
            out["unit"] = "Mrays/s"
            out["what"] = ("the same frames delivered to HOST memory (the image plane::trace_rays returns): PCIe-inclusive, "
                           "never the headline value")
        finally:
            for b in bufs:
                self.free_host_image(b)
        return out

    def synchronize(self) -> int:
        """Waits for the stream; returns C5_OK or C5_RETRY (frame must be rendered again)."""
        return self._check(self.lib.c5_synchronize(self.handle), allow=(C5_RETRY,))

    def stats(self) -> dict:
        st = Stats()
        self._check(self.lib.c5_get_stats(self.handle, C.byref(st)))
        return st.as_dict()

    def walk_kernel_ms(self, reset: bool = True):
        avg, n = C.c_double(), C.c_int64()
        self._check(self.lib.c5_walk_kernel_ms(self.handle, int(reset), C.byref(avg), C.byref(n)))
        return avg.value, n.value

    def view_points(self, n_pts: int) -> np.ndarray:
        out = np.empty((n_pts, 3), dtype=np.float64)
        self._check(self.lib.c5_download_view_points(self.handle, _dp(out)), allow=(C5_RETRY,))
        return out


def face_adjacency(cells, n_pts: int):
    """Host-only: (adj[n,4], n_boundary_faces) via c5_face_adjacency."""
    lib = load_library()
    cells = np.ascontiguousarray(cells, dtype=np.int32).reshape(-1, 4)
    adj = np.empty_like(cells)
    nb = C.c_int64()
    rc = lib.c5_face_adjacency(cells.ctypes.data_as(C.POINTER(C.c_int32)), cells.shape[0], n_pts,
                               adj.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nb))
    if rc != C5_OK:
        raise C5Error(rc, lib.c5_last_error(None).decode())
    return adj, nb.value


def weld_points(xyz):
    """Host-only: (rep[n_pts], n_merged) via c5_weld_points."""
    lib = load_library()
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
    rep = np.empty(xyz.shape[0], dtype=np.int32)
    m = C.c_int64()
    rc = lib.c5_weld_points(_dp(xyz), xyz.shape[0], rep.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(m))
    if rc != C5_OK:
        raise C5Error(rc, lib.c5_last_error(None).decode())
    return rep, m.value


def device_count() -> int:
    lib = load_library()
    n = C.c_int()
    lib.c5_device_count(C.byref(n))
    return n.value
