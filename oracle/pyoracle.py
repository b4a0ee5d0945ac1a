"""TEST INFRASTRUCTURE — ctypes bindings for the CPU oracle and the reference-backed checker.

`Oracle("port")`      -> oracle/_build/liboracle.so   (own restatement, OpenMP)
`Oracle("reference")` -> oracle/_ref/libcourse5_ref.so (real reference line.cpp/tetra.cpp, serial)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PORT_SO = os.path.join(HERE, "_build", "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libcourse5_ref.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)
_fp = C.POINTER(C.c_float)


def build(quiet: bool = True) -> None:
    """Compile the oracle (and the reference-backed checker when /root/reference exists)."""
    subprocess.run(["make", "-C", HERE] + (["-s"] if quiet else []), check=True)


def reference_available() -> bool:
    return os.path.exists(REF_SO)


def _ptr(a, t):
    return a.ctypes.data_as(t) if a is not None else t()


class Oracle:
    def __init__(self, kind: str = "port"):
        self.kind = kind
        path = PORT_SO if kind == "port" else REF_SO
        if not os.path.exists(path):
            if kind == "port":
                build()
            else:
                raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self.prefix = "c5o_" if kind == "port" else "c5r_"
        rot = getattr(self.lib, self.prefix + "rotate_points")
        rot.argtypes = [_dp, C.c_int64, _dp, C.c_int]
        rot.restype = None
        if kind == "port":
            self.lib.c5o_face_z.argtypes = [C.c_double, C.c_double, _dp, _dp, _dp]
            self.lib.c5o_face_z.restype = C.c_double
            self.lib.c5o_emission_step.argtypes = [C.c_double] * 5
            self.lib.c5o_emission_step.restype = C.c_double
            self.lib.c5o_pixel_coords.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp]
            self.lib.c5o_pixel_coords.restype = None

    def rotate_points(self, xyz: np.ndarray, rots: np.ndarray) -> np.ndarray:
        out = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3).copy()
        rots = np.ascontiguousarray(rots, dtype=np.float64).reshape(-1, 3)
        getattr(self.lib, self.prefix + "rotate_points")(_ptr(out, _dp), out.shape[0], _ptr(rots, _dp), rots.shape[0])
        return out.reshape(np.shape(xyz))

    def face_z(self, x, y, a, b, c) -> float:
        a, b, c = (np.ascontiguousarray(v, dtype=np.float64) for v in (a, b, c))
        return self.lib.c5o_face_z(x, y, _ptr(a, _dp), _ptr(b, _dp), _ptr(c, _dp))

    def emission_step(self, I, alpha, Q, dz, alpha_limit) -> float:
        return self.lib.c5o_emission_step(I, alpha, Q, dz, alpha_limit)

    def probe_segments(self, xyz, cells, alpha, q, rots, res_x, res_y, bounds, probes, cap=512):
        """("reference" only) per probed pixel (col, row): [(tetra id, delta z)] exactly as the reference's
        line::calculate_intersections leaves line::_intersections_delta (line.cpp:84-148) — oracle/ref_inspect.cpp."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        cells = np.ascontiguousarray(cells, dtype=np.int32).reshape(-1, 4)
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        q = np.ascontiguousarray(q, dtype=np.float64)
        rots = np.ascontiguousarray(rots, dtype=np.float64).reshape(-1, 3)
        b = np.ascontiguousarray(bounds, dtype=np.float64)
        pij = np.ascontiguousarray(probes, dtype=np.int32).reshape(-1, 2)
        out = np.zeros((len(pij), cap, 2))
        cnt = np.zeros(len(pij), dtype=np.int32)
        err = C.create_string_buffer(512)
        fn = self.lib.c5r_probe
        fn.restype = C.c_int
        rc = fn(_ptr(xyz, _dp), C.c_int64(len(xyz)), _ptr(cells, _ip), C.c_int64(len(cells)), _ptr(alpha, _dp), _ptr(q, _dp),
                _ptr(rots, _dp), C.c_int(len(rots)), C.c_int(res_x), C.c_int(res_y), _ptr(b, _dp), _ptr(pij, _ip),
                C.c_int(len(pij)), C.c_int(cap), _ptr(out, _dp), _ptr(cnt, _ip), err, C.c_int(512))
        if rc != 0:
            raise RuntimeError(err.value.decode())
        return [out[k, :min(cnt[k], cap)].copy() for k in range(len(pij))]

    def scan_face(self, res_x, res_y, bounds, v0, v1, v2, cap=1 << 16):
        """Pixels (col, row) one projected face covers (plane.cpp:57-142), in emission order ("port" only)."""
        b = np.ascontiguousarray(bounds, dtype=np.float64)
        vs = [np.ascontiguousarray(v, dtype=np.float64) for v in (v0, v1, v2)]
        out = np.zeros((cap, 2), dtype=np.int32)
        fn = self.lib.c5o_scan_face
        fn.restype = C.c_int64
        n = fn(C.c_int(res_x), C.c_int(res_y), _ptr(b, _dp), _ptr(vs[0], _dp), _ptr(vs[1], _dp), _ptr(vs[2], _dp),
               _ptr(out, _ip), C.c_int64(cap))
        return out[:min(n, cap)].copy(), int(n)

    def pixel_coords(self, res_x, res_y, bounds):
        b = np.ascontiguousarray(bounds, dtype=np.float64)
        X = np.empty(res_x)
        Y = np.empty(res_y)
        self.lib.c5o_pixel_coords(res_x, res_y, _ptr(b, _dp), _ptr(X, _dp), _ptr(Y, _dp))
        return X, Y

    def render(self, xyz, cells, alpha, q, rots, res_x, res_y, bounds, alpha_limit=2.5,
               solid_tets=None, solid_colour=None, threads=1, probes=None, probe_cap=512, row_stride=1, row_phase=0):
        """Returns dict(image[Y,X,2] float32, segments, covered, marked, timing_ms, probes).
        row_stride > 1 ("port" only): render only the rows j with j % row_stride == row_phase (the others stay
        zero, the counts cover the kept rows) — pixels are independent, so those rows equal a full render's."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        cells = np.ascontiguousarray(cells, dtype=np.int32).reshape(-1, 4)
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        q = np.ascontiguousarray(q, dtype=np.float64)
        rots = np.ascontiguousarray(rots, dtype=np.float64).reshape(-1, 3)
        bounds = np.ascontiguousarray(bounds, dtype=np.float64)
        n_solid = 0
        if solid_tets is not None and len(solid_tets):
            solid_tets = np.ascontiguousarray(solid_tets, dtype=np.float64).reshape(-1, 12)
            n_solid = solid_tets.shape[0]
            solid_colour = np.ascontiguousarray(
                np.broadcast_to(np.asarray(solid_colour, dtype=np.float64), (n_solid,)))
        else:
            solid_tets = solid_colour = None
        out = np.zeros((res_y, res_x, 2), dtype=np.float32)
        stats = np.zeros(4, dtype=np.int64)
        err = C.create_string_buffer(512)
        res = dict()
        if self.kind == "port":
            timing = np.zeros(3)
            n_probe = 0
            pij = pout = pcnt = None
            if probes is not None and len(probes):
                pij = np.ascontiguousarray(probes, dtype=np.int32).reshape(-1, 2)
                n_probe = pij.shape[0]
                pout = np.zeros((n_probe, probe_cap, 3))
                pcnt = np.zeros(n_probe, dtype=np.int32)
            fn = self.lib.c5o_render_rows
            fn.restype = C.c_int
            rc = fn(_ptr(xyz, _dp), C.c_int64(xyz.shape[0]), _ptr(cells, _ip), C.c_int64(cells.shape[0]),
                    _ptr(alpha, _dp), _ptr(q, _dp), _ptr(rots, _dp), C.c_int(rots.shape[0]),
                    _ptr(solid_tets, _dp), _ptr(solid_colour, _dp), C.c_int64(n_solid),
                    C.c_int(res_x), C.c_int(res_y), _ptr(bounds, _dp), C.c_double(alpha_limit),
                    C.c_int(threads), _ptr(out, _fp), _ptr(stats, _lp), _ptr(timing, _dp),
                    _ptr(pij, _ip), C.c_int(n_probe), C.c_int(probe_cap), _ptr(pout, _dp),
                    _ptr(pcnt, _ip), err, C.c_int(512), C.c_int(row_stride), C.c_int(row_phase))
            res["timing_ms"] = timing
            if n_probe:
                res["probes"] = [pout[k, :min(pcnt[k], probe_cap)].copy() for k in range(n_probe)]
        else:
            if row_stride != 1:
                raise ValueError("row_stride is a feature of the port oracle")
            fn = self.lib.c5r_render
            fn.restype = C.c_int
            rc = fn(_ptr(xyz, _dp), C.c_int64(xyz.shape[0]), _ptr(cells, _ip), C.c_int64(cells.shape[0]),
                    _ptr(alpha, _dp), _ptr(q, _dp), _ptr(rots, _dp), C.c_int(rots.shape[0]),
                    _ptr(solid_tets, _dp), _ptr(solid_colour, _dp), C.c_int64(n_solid),
                    C.c_int(res_x), C.c_int(res_y), _ptr(bounds, _dp), C.c_double(alpha_limit),
                    _ptr(out, _fp), _ptr(stats, _lp), err, C.c_int(512))
        if rc != 0:
            raise RuntimeError(err.value.decode())
        res.update(image=out, segments=int(stats[0]), covered=int(stats[1]), marked=int(stats[2]))
        return res
