"""TEST INFRASTRUCTURE — a second, independent restatement (numpy) of the reference's procedural solids, written
from the reference text alone, to pin the product's C++ restatement (course5_amd/csrc/host/scene.cpp) point
for point.  object3d_base.cpp needs VTK headers and cannot be compiled in this image, so the reference's own
object code is out of reach for this part; two restatements by different routes (a
scalar C++ one and a vectorised numpy one) agreeing to the last bit is the next best pin.

Follows /root/reference/project/src:
  object3d_base.cpp:55-81    rotate_vector_around_y_axis / _z_axis, add_vector
  object3d_base.cpp:83-196   init_polar: march along rays until potential >= level; accumulated angles; centre-fan cells
  object3d_roche_lobe.cpp:20-49, object3d_sphere.cpp:11-17, config.hpp:45-72
"""
import ctypes
import math

import numpy as np

# g++ -O3 (the reference's CMakeLists.txt:6-9, and the product's build) merges the cos(a) and sin(a) of one
# angle into ONE call of glibc's sincos(), whose results differ from cos() / sin() in the last bit for about
# one angle in a thousand.  That bit decides on which side of the level set a marched point ends, so the
# restatement calls sincos too.
_libm = ctypes.CDLL("libm.so.6")
_libm.sincos.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
_libm.sincos.restype = None


def _sincos(a):
    s, c = ctypes.c_double(), ctypes.c_double()
    _libm.sincos(a, ctypes.byref(s), ctypes.byref(c))
    return s.value, c.value

PI = 3.14159265358979323846       # config.hpp:45
L, ACC_X0, ACC_DISK_R = 0.945, 1.0, 0.02
M_ACC, M_DONOR = 0.73, 0.1
OMEGA = 2 * PI * 10000
G_SOL = np.longdouble(132700000000000000000.0)  # long double in the reference (config.hpp:69)
EPS = np.finfo(np.float64).eps


def _angles(angle_step):
    """object3d_base.cpp:89-93,123-143: both angles are running sums; the loop bounds decide the ring sizes."""
    d = PI / angle_step
    ys, a = [], -PI + d
    while a < (PI - d + EPS):
        ys.append(a)
        a += d
    xs, a = [], 0.0
    while a < 2 * PI - d + EPS:
        xs.append(a)
        a += d
    return ys, xs


def _march(potential, origin, steps, level):
    """object3d_base.cpp:101-107,133-136 for many rays at once: p += step; until potential(p) >= level.
    steps: [n, 3].  Every ray repeats the same additions the scalar loop makes."""
    p = np.tile(np.asarray(origin, dtype=np.float64), (len(steps), 1))
    live = np.ones(len(steps), dtype=bool)
    while live.any():
        p[live] = p[live] + steps[live]
        live[live] = potential(p[live]) < level
    return p


def init_polar(potential, x0, y0, z0, level, step, angle_step):
    """Returns (cells [n, 4, 3], ring points [rings, per ring, 3], top, bottom)."""
    ys, xs = _angles(angle_step)
    top = _march(potential, (x0, y0, z0), np.array([[0.0, 0.0, step]]), level)[0]
    bottom = _march(potential, (x0, y0, z0), np.array([[0.0, 0.0, -step]]), level)[0]
    steps = np.empty((len(ys), len(xs), 3))
    for i, ay in enumerate(ys):
        # rotate_vector_around_z_axis({0.001, 0, 0}, ay), object3d_base.cpp:67-75
        sy, cy = _sincos(ay)
        v = (0.001 * cy + 0.0 * sy, -0.001 * sy + 0.0 * cy, 0.0)
        for j, ax in enumerate(xs):
            # rotate_vector_around_y_axis(v, ax), object3d_base.cpp:55-65
            sx, cx = _sincos(ax)
            steps[i, j] = (v[0] * cx + v[2] * sx, v[1], -v[0] * sx + v[2] * cx)
    ring = _march(potential, (x0, y0, z0), steps.reshape(-1, 3), level).reshape(len(ys), len(xs), 3)
    centre = np.array([x0, y0, z0])
    n, last = len(xs), len(ys) - 1
    cells = []
    for i in range(1, n):                                            # object3d_base.cpp:158-161
        cells.append((centre, bottom, ring[0][i], ring[0][i - 1]))
    cells.append((centre, bottom, ring[0][0], ring[0][n - 1]))       # :162-163
    for i in range(1, n):                                            # :166-170
        cells.append((centre, top, ring[last][i], ring[last][i - 1]))
    cells.append((centre, top, ring[last][0], ring[0][n - 1]))       # :172-175 (sic: ring 0)
    for i in range(1, last + 1):                                     # :177-193
        for j in range(1, n):
            cells.append((centre, ring[i - 1][j - 1], ring[i - 1][j], ring[i][j - 1]))
            cells.append((centre, ring[i][j - 1], ring[i][j], ring[i - 1][j]))
        cells.append((centre, ring[i - 1][n - 1], ring[i - 1][0], ring[i][n - 1]))
        cells.append((centre, ring[i][n - 1], ring[i][0], ring[i - 1][0]))
    return np.array(cells), ring, top, bottom


def _norm(x, y, z):
    """vector_2_norm: sum += i * i over the components in order, then sqrt (object3d_roche_lobe.cpp:3-9)."""
    return np.sqrt((x * x + y * y) + z * z)


def roche_lobe():
    """object3d_roche_lobe.cpp:20-49 with the constants of main.cpp:110 / config.hpp."""
    donor_x = ACC_X0 - L
    mass_centre_x = (donor_x * M_DONOR + ACC_X0 * M_ACC) / (M_ACC + M_DONOR)

    def potential(r):
        r = np.atleast_2d(r)
        acc = _norm(r[:, 0] - ACC_X0, r[:, 1], r[:, 2])
        don = _norm(r[:, 0] - donor_x, r[:, 1], r[:, 2])
        m0, m1, m2 = r[:, 0] - mass_centre_x, r[:, 1], r[:, 2]
        # vector_multiplication(m, {0, omega, 0}), object3d_roche_lobe.cpp:11-18
        c0 = m1 * 0.0 - m2 * OMEGA
        c1 = -(m0 * 0.0) + (m2 * 0.0)
        c2 = m0 * OMEGA - m1 * 0.0
        spin = _norm(c0, c1, c2)
        omega = (1.0 / 2.0) * spin * spin
        # long double through G_SOL, rounded to double on return (object3d_roche_lobe.cpp:42-43)
        F = -((G_SOL * M_ACC) / acc.astype(np.longdouble)) - ((G_SOL * M_DONOR) / don.astype(np.longdouble)) - omega.astype(np.longdouble)
        return F.astype(np.float64)

    level = potential(np.array([[0.35515, 0.0, 0.0]]))[0]  # the L1 override, object3d_roche_lobe.cpp:30
    return init_polar(potential, donor_x, 0.0, 0.0, level, 0.001, 128)


def sphere():
    """object3d_sphere.cpp:11-17 with main.cpp:116's arguments."""
    def potential(p):
        p = np.atleast_2d(p)
        return _norm(p[:, 0] - ACC_X0, p[:, 1] - 0.0, p[:, 2] - 0.0)
    return init_polar(potential, ACC_X0, 0.0, 0.0, ACC_DISK_R, 0.001, 256)
