"""End to end through the `course` binary on the GPU: .vtk in -> .vti out, compared with the CPU
oracle fed the same grid, view and (product-generated) solids."""
import os
import subprocess

import numpy as np
import pytest

from course5_amd import meshgen as mg, vtkio
from parity import assert_images_match

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COURSE = os.path.join(ROOT, "course5_amd", "course")
PI = 3.14159265358979323846


def load_solids(path):
    raw = open(path, "rb").read()
    out, off = [], 0
    while off < len(raw):
        n = int(np.frombuffer(raw, dtype=np.int64, count=1, offset=off)[0])
        out.append(np.frombuffer(raw, dtype=np.float64, count=12 * n, offset=off + 8).reshape(n, 4, 3))
        off += 8 + 96 * n
    return out


@pytest.mark.parametrize("donor", [0.0, 0.25])
def test_course_end_to_end_with_solids(tmp_path, oracle_port, donor):
    xyz, cells, a, q = mg.workload("g2")
    src, dst, dump = tmp_path / "g2.vtk", tmp_path / "out.vti", tmp_path / "solids.bin"
    mg.write_vtk_ascii(str(src), xyz, cells, a, q)
    rx, ry, ax, ay, limit = 600, 450, 0.1, 0.07, 3.0
    r = subprocess.run([COURSE, "-f", str(src), "-d", str(dst), "-j4", "-x", str(rx), "-y", str(ry), "-X", str(ax),
                        "-Y", str(ay), "-D", str(donor), "--alpha_limit", str(limit), "--stats",
                        "--dump_solids", str(dump)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    for line in ("Loading data with VTK lib and other preparations completed in", "Ray-tracing completed in",
                 "Result exported. Calculations completed."):  # main.cpp:119-123,131-135,138
        assert line in r.stdout
    img, info = vtkio.read_vti(str(dst))
    # object2d.cpp:12-13: dims (X, Y, 1), VTK_DOUBLE x 2 "ImageScalars", default origin/spacing
    assert info == dict(name="ImageScalars", type="Float64", components=2, dims=(rx, ry, 1), origin="0 0 0", spacing="1 1 1")
    assert np.array_equal(img, img.astype(np.float32).astype(np.float64), equal_nan=True)  # fp32-exact doubles (plane.cpp:165-166)

    rots = mg.view_rotations(ax, ay)
    lobe, sphere = load_solids(str(dump))
    lobe_rots = np.vstack([[1.0, donor * PI, 1.0], rots])  # object3d_roche_lobe.cpp:48 then main.cpp:112-114
    lobe_view = oracle_port.rotate_points(lobe.reshape(-1, 3), lobe_rots).reshape(-1, 12)
    solids = np.vstack([lobe_view, sphere.reshape(-1, 12)])  # the sphere is not rotated (main.cpp:116)
    ref = oracle_port.render(xyz, cells, a, q, rots, rx, ry, mg.REFERENCE_BOUNDS, alpha_limit=limit,
                             solid_tets=solids, solid_colour=float("nan"), threads=8)
    got = img.astype(np.float32)
    assert ref["marked"] > 1000
    assert np.array_equal(np.isnan(got), np.isnan(ref["image"]))  # NaN mask of lobe + sphere bit-exact
    assert_images_match(got, ref["image"], "course end to end")


def test_course_sweep_keeps_the_grid_resident(tmp_path, oracle_port):
    xyz, cells, a, q = mg.workload("c1")
    src, dst = tmp_path / "c1.vtk", tmp_path / "f.vti"
    mg.write_vtk_binary(str(src), xyz, cells, a, q, v51=True)
    r = subprocess.run([COURSE, "-f", str(src), "-d", str(dst), "-x", "200", "-y", "150", "-X", "0.1", "-Y", "0.07",
                        "--no_solids", "--frames", "3", "--sweep", "Y", "--sweep_step", "0.05"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    q32 = q.astype(np.float32).astype(np.float64)  # the binary writer stores Q as float
    for k in range(3):
        img, _ = vtkio.read_vti(str(tmp_path / f"f_{k:05d}.vti"))
        ay = 0.07
        for _ in range(k):
            ay += 0.05
        ref = oracle_port.render(xyz, cells, a, q32, mg.view_rotations(0.1, ay), 200, 150, mg.REFERENCE_BOUNDS)
        assert_images_match(img.astype(np.float32), ref["image"], f"frame {k}")


def test_course_donor_sweep_moves_only_the_lobe(tmp_path, oracle_port):
    """BASELINE config 5 in small: -D sweep with the grid and both solids resident on the GPU.  -D only
    turns the Roche lobe (main.cpp:110, object3d_roche_lobe.cpp:48): the volume image is the same in
    every frame, the NaN mask moves, and every frame matches the oracle fed the rotated lobe."""
    xyz, cells, a, q = mg.workload("c1")
    src, dst, dump = tmp_path / "c1.vtk", tmp_path / "d.vti", tmp_path / "solids.bin"
    mg.write_vtk_ascii(str(src), xyz, cells, a, q)
    rx, ry, step = 400, 300, 0.25
    r = subprocess.run([COURSE, "-f", str(src), "-d", str(dst), "-x", str(rx), "-y", str(ry), "-X", "0.1", "-Y", "0.07",
                        "--frames", "3", "--sweep", "D", "--sweep_step", str(step), "--dump_solids", str(dump)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    rots = mg.view_rotations(0.1, 0.07)
    lobe, sphere = load_solids(str(dump))
    masks = []
    for k in range(3):
        img, _ = vtkio.read_vti(str(tmp_path / f"d_{k:05d}.vti"))
        donor = 0.0
        for _ in range(k):
            donor += step
        lobe_view = oracle_port.rotate_points(lobe.reshape(-1, 3), np.vstack([[1.0, donor * PI, 1.0], rots])).reshape(-1, 12)
        ref = oracle_port.render(xyz, cells, a, q, rots, rx, ry, mg.REFERENCE_BOUNDS,
                                 solid_tets=np.vstack([lobe_view, sphere.reshape(-1, 12)]), solid_colour=float("nan"),
                                 threads=8)
        got = img.astype(np.float32)
        assert np.array_equal(np.isnan(got), np.isnan(ref["image"])), k
        assert_images_match(got, ref["image"], f"donor frame {k}")
        masks.append(np.isnan(got[..., 0]))
    assert not np.array_equal(masks[0], masks[1]) and not np.array_equal(masks[1], masks[2])
