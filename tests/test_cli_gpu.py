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


def _run(args, timeout=600):
    r = subprocess.run([COURSE] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    return r


@pytest.mark.parametrize("layout", ["tiles", "blocks"])
@pytest.mark.parametrize("exchange", ["host", "p2p", "rccl"])
def test_course_splits_a_frame_by_rows_over_several_contexts(tmp_path, exchange, layout):
    """`course --devices`: one process, one c5_context per listed GPU; rows in cyclic 16-row tiles or in one
    contiguous block per GPU (a single frame has no earlier frame to measure: equal blocks), every tile / block
    delivered at its final offset (host: each GPU copies its rows into the pinned host image; p2p / rccl:
    into the root GPU's image first; with blocks the root renders its own block in place).  The test box has one GPU, so the list names it three times: everything
    runs except the RCCL transport itself, which refuses duplicate devices — the run must say so and fall
    back to peer copies.  Output equals the single-GPU file value for value; 450 rows leave a short last tile."""
    xyz, cells, a, q = mg.workload("c2")
    src = tmp_path / "c2.vtk"
    mg.write_vtk_binary(str(src), xyz, cells, a, q)
    common = ["-f", src, "-x", 600, "-y", 450, "-X", 0.1, "-Y", 0.07, "-D", 0.3, "--raw_vti", "-j4"]
    _run(common + ["-d", tmp_path / "one.vti"])
    r = _run(common + ["-d", tmp_path / "many.vti", "--devices", "0,0,0", "--exchange", exchange, "--split", "rows",
                       "--row_layout", layout, "--stats"])
    one, _ = vtkio.read_vti(str(tmp_path / "one.vti"))
    many, _ = vtkio.read_vti(str(tmp_path / "many.vti"))
    assert np.array_equal(one, many, equal_nan=True)
    assert np.isnan(one).any() and (one[~np.isnan(one)] > 0).any()
    if exchange == "rccl":
        assert "exchanging by peer copies instead" in r.stderr
    segs = [int(w) for line in r.stdout.splitlines() if "segments" in line for w in [line.split(";")[1].split()[0]]]
    assert segs and segs[0] > 0


@pytest.mark.parametrize("exchange", ["host", "p2p"])
def test_course_sweep_in_cost_balanced_row_blocks(tmp_path, exchange):
    """`--split rows` of a sweep: ONE contiguous block of rows per GPU (one message / copy per GPU and
    frame), equal blocks for the first frame, then sized by the segments per row the walk counted (probe frames) —
    the ball sits in the middle of the image, so equal blocks are badly unbalanced and the rows MUST move; frames
    are issued ahead, so the move happens with frames in flight.  Every file equals the single-context one."""
    xyz, cells, a, q = mg.workload("c2")
    src = tmp_path / "c2.vtk"
    mg.write_vtk_binary(str(src), xyz, cells, a, q)
    common = ["-f", src, "-x", 500, "-y", 375, "-X", 0.1, "-Y", 0.07, "-D", 0.2, "--frames", 9, "--sweep", "Y",
              "--sweep_step", 0.04, "--raw_vti", "-j4", "--stats"]
    _run(common + ["-d", tmp_path / "one.vti"])
    r = _run(common + ["-d", tmp_path / "blk.vti", "--devices", "0,0,0", "--split", "rows", "--exchange", exchange])
    for k in range(9):
        one, _ = vtkio.read_vti(str(tmp_path / f"one_{k:05d}.vti"))
        blk, _ = vtkio.read_vti(str(tmp_path / f"blk_{k:05d}.vti"))
        assert np.array_equal(one, blk, equal_nan=True), k
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("Row blocks laid out anew")]
    assert line, r.stdout[-800:]
    moved = int(line[0].split(":")[1].split(";")[0])
    rows = [int(w) for w in line[0].rsplit(":", 1)[1].split()]
    assert moved >= 1 and sum(rows) == 375 and len(rows) == 3
    assert rows[1] < rows[0] and rows[1] < rows[2]  # the block through the ball is the thinnest


def test_course_deals_the_frames_of_a_sweep_to_several_contexts(tmp_path):
    """Frame-parallel sweep (BASELINE config 5 on several GPUs): frame k on context k mod N, each with the
    grid and the solids resident, frames issued ahead and written in order.  Every file equals the one the
    single-context sweep writes."""
    xyz, cells, a, q = mg.workload("g2")
    src = tmp_path / "g2.vtk"
    mg.write_vtk_ascii(str(src), xyz, cells, a, q)
    common = ["-f", src, "-x", 400, "-y", 300, "-X", 0.1, "-Y", 0.07, "--frames", 7, "--sweep", "D", "--sweep_step", 0.11,
              "--raw_vti", "-j4"]
    _run(common + ["-d", tmp_path / "a.vti"])
    _run(common + ["-d", tmp_path / "b.vti", "--devices", "0,0,0"])
    masks = []
    for k in range(7):
        one, _ = vtkio.read_vti(str(tmp_path / f"a_{k:05d}.vti"))
        many, _ = vtkio.read_vti(str(tmp_path / f"b_{k:05d}.vti"))
        assert np.array_equal(one, many, equal_nan=True), k
        masks.append(np.isnan(one[..., 0]))
    assert all(not np.array_equal(masks[0], m) for m in masks[1:])


def test_course_bench_prints_one_json_line(tmp_path):
    import json
    xyz, cells, a, q = mg.workload("g2")
    src = tmp_path / "g2.vtk"
    mg.write_vtk_ascii(str(src), xyz, cells, a, q)
    for extra, split in ((["--devices", "0,0"], "frames"), (["--devices", "0,0", "--split", "rows"], "rows"), ([], "none")):
        r = _run(["-f", src, "-x", 640, "-y", 480, "-X", 0.1, "-Y", 0.07, "--no_solids", "--bench", 30, "--bench_warmup", 5,
                  "--sweep", "Y"] + extra)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('{"course_bench"')]
        assert len(line) == 1
        b = json.loads(line[0])["course_bench"]
        assert b["frames"] == 30 and b["split"] == split and b["mrays_per_s"] > 10 and b["retries"] == 0
        assert b["row_layout"] == ("blocks" if split == "rows" else "none")
        assert not list(tmp_path.glob("*.vti"))
    # --bench_files: the timed frames are written like a sweep's (the end-to-end figure with files)
    r = _run(["-f", src, "-d", tmp_path / "bf.vti", "-x", 640, "-y", 480, "-X", 0.1, "-Y", 0.07, "--no_solids", "--bench", 6,
              "--bench_warmup", 2, "--bench_files", "--sweep", "Y", "-j4"])
    b = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"course_bench"')][0])["course_bench"]
    assert b["delivered_to"] == "zlib .vti files" and b["frames_per_s"] > 1
    assert len(list(tmp_path.glob("bf_*.vti"))) == 6
    img, _ = vtkio.read_vti(str(tmp_path / "bf_00003.vti"))
    assert img.shape == (480, 640, 2) and (img[..., 0] > 0).sum() > 10_000


def test_course_automatic_boundaries(tmp_path, oracle_port):
    """plane.cpp:278-288: without manual boundaries the image domain is the x / y bounding box of the transformed
    objects (object3d_base.cpp:221-255, tetra.cpp:18-42).  The reference's main never takes that path
    (main.cpp:83); `course --auto_bounds` does, and must agree with the oracle fed the same box."""
    xyz, cells, a, q = mg.workload("g2")
    src, dst = tmp_path / "g2.vtk", tmp_path / "auto.vti"
    mg.write_vtk_ascii(str(src), xyz, cells, a, q)
    _run(["-f", src, "-d", dst, "-x", 300, "-y", 200, "-X", 0.1, "-Y", 0.07, "--no_solids", "--auto_bounds", "--raw_vti"])
    rots = mg.view_rotations(0.1, 0.07)
    v = oracle_port.rotate_points(xyz, rots)
    bounds = (v[:, 0].max(), v[:, 0].min(), v[:, 1].max(), v[:, 1].min())
    img, _ = vtkio.read_vti(str(dst))
    # the grid touches all four borders of its own bounding box, where the reference's clamp smears faces in odd
    # numbers (it would abort: plane.cpp:39-41), so compare with the walk's own render in the oracle's domain:
    # every pixel strictly inside must match the oracle rendered in a domain one pixel wider
    sx, sy = (bounds[0] - bounds[1]) / 299, (bounds[2] - bounds[3]) / 199
    wide = (bounds[0] + sx, bounds[1] - sx, bounds[2] + sy, bounds[3] - sy)
    ref = oracle_port.render(xyz, cells, a, q, rots, 302, 202, wide, threads=8)
    got = img.astype(np.float32)
    want = ref["image"][1:-1, 1:-1]
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    bad = d > 1e-5 * np.maximum(np.abs(got), np.abs(want)) + 1e-6 * np.abs(want).max()
    assert bad.sum() <= 8 and (got[..., 0] > 0).sum() > 20_000


def test_course_sweep_survives_a_starved_entry_pool(tmp_path):
    """plane::trace_rays' C5_RETRY handling end to end: a sweep over the non-convex ball started from an overflow
    pool of 100 records (test hook C5_TEST_ENTRY_POOL), frames issued ahead.  The frame that overflows and the
    frames already in flight behind it are rendered again, in order, each with its own view: every file equals
    the one of an undisturbed run, and the run says that it had to render frames again."""
    xyz, cells, a, q = mg.workload("c2")
    src = tmp_path / "c2.vtk"
    mg.write_vtk_binary(str(src), xyz, cells, a, q)
    common = ["-f", src, "-x", 500, "-y", 375, "-X", 0.1, "-Y", 0.07, "--no_solids", "--frames", 6, "--sweep", "Y",
              "--sweep_step", 0.05, "--raw_vti", "--stats", "-j4"]
    _run(common + ["-d", tmp_path / "calm.vti"])
    env = dict(os.environ, C5_TEST_ENTRY_POOL="100")
    r = subprocess.run([COURSE] + [str(x) for x in common + ["-d", tmp_path / "starved.vti"]], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-1000:]
    again = [int(line.rsplit(":", 1)[1]) for line in r.stdout.splitlines() if line.startswith("Frames rendered again")]
    assert again and again[0] >= 1
    for k in range(6):
        calm, _ = vtkio.read_vti(str(tmp_path / f"calm_{k:05d}.vti"))
        starved, _ = vtkio.read_vti(str(tmp_path / f"starved_{k:05d}.vti"))
        assert np.array_equal(calm, starved), k
    # the same with the rows of every frame split over two contexts, tiles and blocks, host copies and peer copies —
    # and with count_all_intersections() called between find_intersections() and trace_rays() (C5_TEST_STATS_BETWEEN:
    # the reference's API allows it, plane.cpp:3-12): the stats call then meets the C5_RETRY first, the library
    # settles it there (pool grown, failure words cleared), and trace_rays must still render the frames again
    # (plane.cpp:541 swallowed it)
    runs = [("0,0", "tiles", "host", False), ("0,0", "blocks", "host", False), ("0,0", "blocks", "host", True),
            ("0,0", "blocks", "p2p", True), ("0,0", "tiles", "p2p", True), ("0", "blocks", "host", True)]
    for n, (devices, layout, exchange, between) in enumerate(runs):
        e2 = dict(env, C5_TEST_STATS_BETWEEN="1") if between else env
        extra = ["--devices", devices, "--split", "rows", "--row_layout", layout, "--exchange", exchange] if devices != "0" else []
        r = subprocess.run([COURSE] + [str(x) for x in common + ["-d", tmp_path / f"r{n}.vti"] + extra],
                           capture_output=True, text=True, timeout=600, env=e2)
        assert r.returncode == 0, (runs[n], r.stderr[-1000:])
        again = [int(line.rsplit(":", 1)[1]) for line in r.stdout.splitlines() if line.startswith("Frames rendered again")]
        assert again and again[0] >= 1, runs[n]
        for k in range(6):
            calm, _ = vtkio.read_vti(str(tmp_path / f"calm_{k:05d}.vti"))
            got, _ = vtkio.read_vti(str(tmp_path / f"r{n}_{k:05d}.vti"))
            assert np.array_equal(calm, got), (runs[n], k)


def test_course_sweep_writes_colour_mapped_frames(tmp_path):
    """f-3: the sweep driver also writes what utility/screen.py makes per frame — a PNG of channel 'Y' in ParaView's
    default map — so that rotate_traces.py's 1500 process launches + pvpython calls become one run.  The PNG of a
    frame is its .vti mapped: NaN (solid) pixels yellow, everything else the colour of its intensity over the
    fixed range, rows top down."""
    xyz, cells, a, q = mg.workload("g2")
    src = tmp_path / "g2.vtk"
    mg.write_vtk_ascii(str(src), xyz, cells, a, q)
    _run(["-f", src, "-d", tmp_path / "f.vti", "-x", 300, "-y", 225, "-X", 0.1, "-Y", 0.07, "--frames", 3, "--sweep", "D",
          "--sweep_step", 0.1, "--png", "--png_range", "0,0.5", "-j4"])
    for k in range(3):
        img, _ = vtkio.read_vti(str(tmp_path / f"f_{k:05d}.vti"))
        rgb = vtkio.read_png(str(tmp_path / f"f_{k:05d}.png"))[::-1].astype(int)  # back to image rows
        assert rgb.shape == (225, 300, 3)
        nan = np.isnan(img[..., 1])
        assert nan.sum() > 100 and (rgb[nan] == np.array([255, 255, 0])).all()
        empty = (~nan) & (img[..., 1] == 0)
        assert empty.any() and (rgb[empty] == np.array([59, 76, 192])).all()      # the cool end
        lit = (~nan) & (img[..., 1] > 0.01)
        assert lit.sum() > 1000 and (rgb[lit] != np.array([59, 76, 192])).any(axis=-1).all()
        # brighter pixels are never bluer: the red component grows with the value up to the middle of the map
        lower_half = (~nan) & (img[..., 1] < 0.25)
        order = np.argsort(img[..., 1][lower_half], kind="stable")
        assert (np.diff(rgb[..., 0][lower_half][order]) >= 0).all()


def test_rccl_entry_points_and_call_pattern_on_one_gpu():
    """What one GPU can prove of `--exchange rccl` (the 8-GPU exchange itself only runs on a real node): librccl is
    loaded on demand, every entry point the exchange uses resolves, a communicator comes up, and the exchange's
    call pattern — one group of ncclSend / ncclRecv pairs landing 16-row tiles (the last one short) at their
    final offsets — moves exactly the right floats, the device sending to itself."""
    r = subprocess.run([COURSE, "--rccl_selftest"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    assert "RCCL self-test ok" in r.stdout and "5 grouped ncclSend/ncclRecv pairs" in r.stdout
    assert "2 pairs (one per peer)" in r.stdout  # the blocks layout: one message per GPU and frame
