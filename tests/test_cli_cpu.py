"""`course` command line: option table, banner, input formats, procedural solids.  No GPU needed
(--parse_only stops before any device work)."""
import os
import re
import subprocess

import numpy as np
import pytest

from course5_amd import meshgen as mg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COURSE = os.path.join(ROOT, "course5_amd", "course")

pytestmark = pytest.mark.skipif(not os.path.exists(COURSE), reason="course CLI not built (run __graft_entry__.build())")


def run(*args):
    return subprocess.run([COURSE, *args], capture_output=True, text=True, timeout=300)


def test_help_lists_the_reference_options():
    r = run("--help")
    assert r.returncode == 0 and r.stdout.startswith("Allowed options:")
    # readme.md:20-36 / main.cpp:21-33: names, short forms and defaults
    for line in ["-h [ --help ]", "-f [ --file ] arg", "-d [ --destination ] arg", "-j [ --threads ] arg",
                 "-x [ --resolution_x ] arg (=1200)", "-y [ --resolution_y ] arg (=900)",
                 "-X [ --angle_around_x ] arg (=0)", "-Y [ --angle_around_y ] arg (=0)",
                 "-D [ --donor_angle ] arg (=0)", "-I [ --initial_system_angle ] arg (=0)",
                 "--alpha_limit arg (=2.5)"]:
        assert line in r.stdout, line


def test_missing_file_or_destination_prints_the_message_and_returns_zero():
    r = run("-f", "only_source.vtk")
    assert r.returncode == 0  # main.cpp:76-78
    assert r.stdout.startswith("Error! Source filename and destination filename must be specified")


def test_unknown_option_fails():
    r = run("--no_such_option")
    assert r.returncode != 0 and "unrecognised option" in r.stderr


@pytest.fixture(scope="module")
def grids(tmp_path_factory):
    d = tmp_path_factory.mktemp("vtk")
    xyz, cells, a, q = mg.workload("g2")
    mg.write_vtk_ascii(str(d / "ascii.vtk"), xyz, cells, a, q)
    mg.write_vtk_binary(str(d / "binary.vtk"), xyz, cells, a, q)
    mg.write_vtk_binary(str(d / "v51.vtk"), xyz, cells, a, q, v51=True)
    return d, a, q


@pytest.mark.parametrize("name", ["ascii.vtk", "binary.vtk", "v51.vtk"])
def test_reader_handles_ascii_binary_and_v51_layouts(grids, name):
    d, a, q = grids
    r = run("-f", str(d / name), "-d", str(d / "o.vti"), "-j16", "-x", "64", "--resolution_y=48", "-X", "0.5",
            "--alpha_limit", "3.0", "--parse_only", "--no_solids")
    assert r.returncode == 0, r.stderr
    # banner (main.cpp:85-92)
    assert "Defined grid resolution: 64x48" in r.stdout
    assert "Number of parallel threads: 16" in r.stdout
    assert "Plane angle around x: 0.5 Pi" in r.stdout
    assert "Limit alpha value: 3" in r.stdout
    m = re.search(r"grid: (\d+) points, (\d+) cells, sum\(alpha\) = (\S+), sum\(Q\) = (\S+)", r.stdout)
    assert (int(m.group(1)), int(m.group(2))) == (125, 384)
    assert float(m.group(3)) == pytest.approx(a.sum(), rel=1e-14)
    q_expected = q if name == "ascii.vtk" else q.astype(np.float32).astype(np.float64)  # binary files store Q as float
    assert float(m.group(4)) == pytest.approx(q_expected.sum(), rel=1e-12)


def test_procedural_solids_have_the_surveyed_cell_counts(grids):
    """SURVEY.md §2 rows 7-8 [probe]: the complete reference produced 130 560 lobe cells and
    522 242 sphere cells; the accumulated angle loops of init_polar decide these numbers."""
    d, _, _ = grids
    dump = d / "solids.bin"
    r = run("-f", str(d / "ascii.vtk"), "-d", str(d / "o.vti"), "--parse_only", "--dump_solids", str(dump))
    assert r.returncode == 0, r.stderr
    assert "roche lobe: 130560 solid cells" in r.stdout and "sphere: 522242 solid cells" in r.stdout
    raw = np.fromfile(dump, dtype=np.uint8)
    n_lobe = int(np.frombuffer(raw[:8].tobytes(), dtype=np.int64)[0])
    lobe = np.frombuffer(raw[8:8 + n_lobe * 96].tobytes(), dtype=np.float64).reshape(n_lobe, 4, 3)
    off = 8 + n_lobe * 96
    n_sph = int(np.frombuffer(raw[off:off + 8].tobytes(), dtype=np.int64)[0])
    sph = np.frombuffer(raw[off + 8:off + 8 + n_sph * 96].tobytes(), dtype=np.float64).reshape(n_sph, 4, 3)
    assert (n_lobe, n_sph) == (130560, 522242)
    # sphere: every tet is a fan from the accretor centre to the R = 0.02 surface, marched in 0.001 steps
    assert np.allclose(sph[:, 0], [1.0, 0.0, 0.0])
    rad = np.linalg.norm(sph[:, 1:] - [1.0, 0.0, 0.0], axis=2)
    assert rad.min() >= 0.02 - 1e-12 and rad.max() <= 0.021 + 1e-9
    # lobe: fan from the donor at x = 1 - 0.945 to the equipotential through the hard-coded point
    # x = 0.35515 (object3d_roche_lobe.cpp:30,44): every surface point is the first 0.001-step along its
    # ray whose potential reaches that level
    donor = np.array([1.0 - 0.945, 0.0, 0.0])
    assert np.allclose(lobe[:, 0], donor)

    def potential(p):  # object3d_roche_lobe.cpp:32-44 in float64 (the product uses long double for G_SOL)
        g, m_a, m_d, omega = 132700000000000000000.0, 0.73, 0.1, 2 * 3.14159265358979323846 * 10000
        mc = (donor[0] * m_d + 1.0 * m_a) / (m_a + m_d)
        r_a = np.linalg.norm(p - [1.0, 0, 0], axis=-1)
        r_d = np.linalg.norm(p - donor, axis=-1)
        spin = omega * np.hypot(p[..., 0] - mc, p[..., 2])
        return -(g * m_a) / r_a - (g * m_d) / r_d - 0.5 * spin * spin

    level = potential(np.array([0.35515, 0.0, 0.0]))
    surf = lobe[::7, 1:].reshape(-1, 3)
    surf = surf[np.linalg.norm(surf - donor, axis=1) > 1e-9]
    direction = (surf - donor) / np.linalg.norm(surf - donor, axis=1, keepdims=True)
    reached = potential(surf) >= level * (1 + 1e-12)
    one_step_before = potential(surf - 0.001 * direction) < level * (1 - 1e-12)
    assert reached.mean() > 0.999 and one_step_before.mean() > 0.999
    assert -0.08 < lobe[..., 0].min() < -0.07 and 0.26 < lobe[..., 0].max() < 0.28


def test_input_errors_are_reported(tmp_path):
    r = run("-f", str(tmp_path / "missing.vtk"), "-d", str(tmp_path / "o.vti"), "--parse_only", "--no_solids")
    assert r.returncode == 1 and "cannot open" in r.stderr
    xyz, cells, a, q = mg.workload("c1")
    p = tmp_path / "no_q.vtk"
    mg.write_vtk_ascii(str(p), xyz, cells, a, q)
    p.write_text(p.read_text().replace("radEnLooseRate", "somethingElse"))
    r = run("-f", str(p), "-d", str(tmp_path / "o.vti"), "--parse_only", "--no_solids")
    assert r.returncode == 1 and "radEnLooseRate" in r.stderr


@pytest.mark.parametrize("raw", [False, True], ids=["zlib-blocks", "raw-appended"])
def test_vti_writer_round_trip(tmp_path, raw):
    """Both encodings of the .vti writer decode to the same Float64 x 2 "ImageScalars" image
    (object2d.cpp:12-21); the default is vtkZLibDataCompressor blocks like vtkXMLImageDataWriter."""
    from course5_amd import vtkio
    out = tmp_path / "t.vti"
    r = run("--selftest_vti", str(out), *( ["--raw_vti"] if raw else []))
    assert r.returncode == 0, r.stderr
    img, info = vtkio.read_vti(str(out))
    assert info["name"] == "ImageScalars" and info["type"] == "Float64" and info["components"] == 2
    assert info["dims"] == (48, 32, 1) and info["origin"] == "0 0 0" and info["spacing"] == "1 1 1"
    yy, xx = np.mgrid[0:32, 0:48]
    want = np.stack([xx + 100.0 * yy, xx + 100.0 * yy + 0.5], axis=-1)
    want[5, 7] = np.nan
    assert np.array_equal(img, want, equal_nan=True)
    head = out.read_bytes()[:600].decode(errors="replace")
    assert ("vtkZLibDataCompressor" in head) == (not raw)


# K. Moreland's published 33-entry table of the "Cool to Warm" map (CoolWarmUChar33), scalar k / 32 -> RGB
COOL_TO_WARM_33 = [(59, 76, 192), (68, 90, 204), (77, 104, 215), (87, 117, 225), (98, 130, 234), (108, 142, 241),
                   (119, 154, 247), (130, 165, 251), (141, 176, 254), (152, 185, 255), (163, 194, 255), (174, 201, 253),
                   (184, 208, 249), (194, 213, 244), (204, 217, 238), (213, 219, 230), (221, 221, 221), (229, 216, 209),
                   (236, 211, 197), (241, 204, 185), (245, 196, 173), (247, 187, 160), (247, 177, 148), (247, 166, 135),
                   (244, 154, 123), (241, 141, 111), (236, 127, 99), (229, 112, 88), (222, 96, 77), (213, 80, 66),
                   (203, 62, 56), (192, 40, 47), (180, 4, 38)]


def test_png_writer_colour_map_orientation_and_nan(tmp_path):
    """`--png`: what utility/screen.py gets from ParaView per frame (component 'Y' coloured with the default map),
    written by the binary itself.  The map is checked against the published 33-entry table of Cool to Warm (the
    256-entry table of the writer may round a component one step differently), NaN pixels are ParaView's yellow, the
    top scanline is the image's last row (y up), and without --png_range a frame is mapped over its own finite range."""
    from course5_amd import vtkio
    out = tmp_path / "t.vti"
    r = run("--selftest_vti", str(out), "--png", "--png_channel", "0", "--png_range", "0,32")
    assert r.returncode == 0, r.stderr
    rgb = vtkio.read_png(str(tmp_path / "t.png")).astype(int)
    assert rgb.shape == (32, 48, 3)
    bottom = rgb[-1]  # image row 0: values 0 ... 47
    assert np.abs(bottom[:33] - np.array(COOL_TO_WARM_33)).max() <= 1
    assert (bottom[33:] == np.array(COOL_TO_WARM_33[-1])).all()      # above the range: clamped to the warm end
    assert tuple(rgb[31 - 5, 7]) == (255, 255, 0)                     # the NaN pixel (row 5, column 7)
    assert (rgb[0] == np.array(COOL_TO_WARM_33[-1])).all()            # image row 31: values 3100 ...

    r = run("--selftest_vti", str(out), "--png")  # channel 1 over its own range 0.5 ... 3147.5
    assert r.returncode == 0, r.stderr
    rgb = vtkio.read_png(str(tmp_path / "t.png")).astype(int)
    assert tuple(rgb[-1, 0]) == COOL_TO_WARM_33[0] and tuple(rgb[0, -1]) == COOL_TO_WARM_33[-1]
    img, _ = vtkio.read_vti(str(out))
    v = img[..., 1]
    t = (v - np.nanmin(v)) / (np.nanmax(v) - np.nanmin(v))
    grey = np.abs(t - 0.5) < 0.01                                    # the middle of the range is the map's grey
    assert grey.any() and (np.abs(rgb[::-1][grey] - 221) <= 4).all()


def test_png_options_are_validated():
    assert run("--selftest_vti", "/dev/null", "--png_channel", "2").returncode == 1
    assert run("--selftest_vti", "/dev/null", "--png_range", "5").returncode == 1
