"""The CPU side under AddressSanitizer + UndefinedBehaviorSanitizer: the host-only
translation units of the product — the hand-written .vtk tokenizer / big-endian / v5.1 reader and the zlib / base64 /
PNG writers (csrc/host/vtk_io.cpp), the option parser (cli.cpp), the procedural solids (scene.cpp), welding + face
adjacency + unique solid faces (csrc/adjacency.cpp) — and the oracle (oracle/oracle.cpp), built by g++ with
-fsanitize=address,undefined into one program (tests/cpp/host_san_main.cpp: no HIP runtime, no GPU).

Every run, on well-formed and on truncated / mis-sized / negative-id / wrong-endian / garbage input, must end in a
result or in an error MESSAGE (exit code 1, "host_san: ..."), never in a sanitizer report.  Build container only
(the GPU pool refuses GPU sanitizers; this is CPU code)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from course5_amd import meshgen as mg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["tests/cpp/host_san_main.cpp", "course5_amd/csrc/host/vtk_io.cpp", "course5_amd/csrc/host/fast_deflate.cpp", "course5_amd/csrc/host/cli.cpp",
           "course5_amd/csrc/host/scene.cpp", "course5_amd/csrc/host/row_blocks.cpp", "course5_amd/csrc/adjacency.cpp", "oracle/oracle.cpp"]

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")


@pytest.fixture(scope="module")
def san(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("san") / "host_san")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined,float-cast-overflow,float-divide-by-zero",
           "-fno-sanitize-recover=undefined", "-fopenmp", "-D__HIP_PLATFORM_AMD__", "-I", "include", "-I", "/opt/rocm/include",
           "-I", "course5_amd/csrc", "-I", "course5_amd/csrc/host"] + SOURCES + ["-o", exe, "-lz"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        pytest.skip("sanitizer build failed here: " + r.stderr[-400:])

    def run(*args, **extra_env):
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="4",
                   **{k: str(v) for k, v in extra_env.items()})
        p = subprocess.run([exe, *[str(a) for a in args]], capture_output=True, text=True, errors="replace", timeout=120, env=env)
        report = "Sanitizer" in p.stderr or "runtime error" in p.stderr or "ERROR: " in p.stderr
        assert not report, (args, p.stderr[-3000:])
        assert p.returncode in (0, 1), (args, p.returncode, p.stderr[-2000:])  # 1 = error message, anything else = crash
        if p.returncode == 1:
            assert p.stderr.startswith("host_san: "), p.stderr[-500:]
        return p
    return run


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("vtk")
    xyz, cells, a, q = mg.workload("g2")
    mg.write_vtk_ascii(str(d / "ascii.vtk"), xyz, cells, a, q)
    mg.write_vtk_binary(str(d / "binary.vtk"), xyz, cells, a, q)
    mg.write_vtk_binary(str(d / "v51.vtk"), xyz, cells, a, q, v51=True)
    xr, cr, _ = mg.refined_interface(3, 1, 2)
    mg.write_vtk_ascii(str(d / "refined.vtk"), xr, cr, *mg.scalars(len(cr)))
    xs, cs = mg.per_cell_point_copies(xyz, cells)
    mg.write_vtk_binary(str(d / "soup.vtk"), xs, cs, a, q, v51=True)
    return d


def test_well_formed_inputs(san, files):
    for name, merged in (("ascii.vtk", 0), ("binary.vtk", 0), ("v51.vtk", 0), ("refined.vtk", 0), ("soup.vtk", 1536 - 125)):
        p = san("read", files / name)
        assert p.returncode == 0, p.stderr
        assert f"merged {merged} " in p.stdout and "conforming 1" in p.stdout, p.stdout
    assert "points 125 cells 384 scalars 2" in san("read", files / "binary.vtk").stdout


def test_malformed_files_end_in_a_message(san, files, tmp_path):
    """Truncations at every structural boundary and in the middle of arrays, counts that lie, ids out of range and
    negative, the wrong byte order, binary garbage, empty and missing files."""
    rng = np.random.default_rng(5)
    n_err = 0
    for name in ("ascii.vtk", "binary.vtk", "v51.vtk"):
        raw = (files / name).read_bytes()
        cases = {}
        for frac in (0.0, 0.01, 0.05, 0.2, 0.37, 0.5, 0.61, 0.8, 0.93, 0.99, 0.999):
            cases[f"cut{frac}"] = raw[: int(len(raw) * frac)]
        for key in (b"POINTS", b"CELLS", b"CELL_TYPES", b"CELL_DATA", b"SCALARS", b"LOOKUP_TABLE", b"OFFSETS", b"CONNECTIVITY", b"FIELD"):
            at = raw.find(key)
            if at >= 0:
                cases[f"cut_at_{key.decode()}"] = raw[:at + len(key)]
                cases[f"cut_after_{key.decode()}"] = raw[:at + len(key) + 9]
        cases["points_count_huge"] = raw.replace(b"POINTS 125", b"POINTS 999999999", 1)
        cases["points_count_small"] = raw.replace(b"POINTS 125", b"POINTS 5", 1)
        cases["points_count_negative"] = raw.replace(b"POINTS 125", b"POINTS -125", 1)
        cases["cells_count_huge"] = raw.replace(b"CELLS 384", b"CELLS 2000000000", 1)
        cases["cells_count_lies"] = raw.replace(b"CELLS 384 1920", b"CELLS 384 19", 1)
        cases["cell_data_count_lies"] = raw.replace(b"CELL_DATA 384", b"CELL_DATA 383", 1)
        cases["type_unknown"] = raw.replace(b" double", b" quadruple", 1)
        cases["no_dataset"] = raw.replace(b"DATASET UNSTRUCTURED_GRID", b"DATASET POLYDATA", 1)
        cases["binary_says_ascii"] = raw.replace(b"BINARY", b"ASCII", 1)
        cases["ascii_says_binary"] = raw.replace(b"ASCII", b"BINARY", 1)
        for k in range(6):  # random byte damage
            b = bytearray(raw)
            for _ in range(40):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            cases[f"noise{k}"] = bytes(b)
        if name == "ascii.vtk":
            text = raw.decode()
            head, rest = text.split("CELLS 384 1920\n", 1)
            lines = rest.split("\n")
            for label, row in (("negative_id", "4 0 1 -7 3"), ("id_out_of_range", "4 0 1 2 125"), ("huge_id", "4 0 1 2 99999999999999"),
                               ("short_cell", "3 0 1 2"), ("word", "4 0 one 2 3")):
                cases[label] = (head + "CELLS 384 1920\n" + "\n".join([row] + lines[1:])).encode()
            cases["nan_coordinate"] = raw.replace(raw.split(b"\n")[5], b"nan 0 0", 1)
        for label, data in cases.items():
            f = tmp_path / f"{name}.{label}"
            f.write_bytes(data)
            p = san("read", f)
            n_err += p.returncode == 1
    san("read", tmp_path / "does_not_exist.vtk")
    (tmp_path / "empty.vtk").write_bytes(b"")
    assert san("read", tmp_path / "empty.vtk").returncode == 1
    (tmp_path / "garbage.vtk").write_bytes(rng.integers(0, 256, 5000, dtype=np.uint8).tobytes())
    assert san("read", tmp_path / "garbage.vtk").returncode == 1
    assert n_err > 60  # most of the damage is noticed and reported (some byte noise only changes values)


def test_writers_and_parser_and_solids(san, tmp_path):
    from course5_amd import vtkio
    for w, h, raw in ((300, 200, 0), (37, 19, 1), (2, 2, 0), (4096, 3, 0), (1, 4100, 0)):
        out = tmp_path / f"t{w}x{h}_{raw}.vti"
        assert san("vti", out, w, h, raw).returncode == 0
        img, info = vtkio.read_vti(str(out))
        assert img.shape == (h, w, 2) and info["dims"] == (w, h, 1)
        assert np.isnan(img[-1, -1, 1]) and np.signbit(img[0, 0, 0]) and img[0, 0, 0] == 0
        assert vtkio.read_png(str(out) + ".png").shape == (h, w, 3)
    assert "go 1" in san("cli", "-f", "a.vtk", "-d", "b.vti", "-j16", "--frames", "3", "--devices", "0-7", "--row_layout", "blocks").stdout
    for bad in (["-x"], ["--resolution_x", "12z"], ["--png_range", "1"], ["--nope"], ["-f"], ["--row_layout", "diagonal", "-f", "a", "-d", "b"],
                ["positional"], ["--split=sideways", "-f", "a", "-d", "b"], ["--png_channel", "7"]):
        assert san("cli", *bad).returncode == 1, bad
    p = san("solids")
    assert "solid cells 130560" in p.stdout and "solid cells 522242" in p.stdout


def test_fast_deflate_round_trips_through_zlib(san):
    """csrc/host/fast_deflate.cpp (the .vti writer's encoder for doubles widened from floats: a fixed parse, one dynamic
    Huffman block) against zlib's own inflate: 1 500 blocks of every kind of value - noisy, smooth, mostly zero, runs, random
    bit patterns with NaNs / denormals / infinities, signed zeros - and sizes from 2 to 4 096 values must come back byte for
    byte; doubles that are not widened floats and buffers that are too small must be declined without a byte written
    past the end."""
    p = san("deflate", 1500)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "deflate ok" in p.stdout


def test_oracle_under_the_sanitizers(san, tmp_path, oracle_port):
    """The checker itself: the sanitized build renders the same bits as the optimised one, with solids, on a grid
    with hanging nodes, and reports a degenerate alignment as a message."""
    xyz, cells, _ = mg.refined_interface(3, 1, 2, jitter=0.1, warp=0.05)
    a, q = mg.scalars(len(cells), seed=8)
    rots = mg.view_rotations(0.1, 0.07)
    solid = (np.array([[[0.9, -0.1, 0.3], [1.1, -0.1, 0.3], [1.0, 0.1, 0.3], [1.0, 0.0, 0.5]]]) + 0.0).reshape(1, 12)

    def dump(path, rx, ry, rot):
        with open(path, "wb") as f:
            f.write(np.array([len(xyz), len(cells), len(rot), rx, ry, 1], dtype=np.int64).tobytes())
            for arr, t in ((xyz, np.float64), (cells, np.int32), (a, np.float64), (q, np.float64), (rot, np.float64),
                           (np.array(mg.REFERENCE_BOUNDS), np.float64), (np.array([2.5]), np.float64), (solid, np.float64),
                           (np.array([np.nan]), np.float64)):
                f.write(np.ascontiguousarray(arr, dtype=t).tobytes())

    dump(tmp_path / "scene.bin", 160, 120, rots)
    p = san("oracle", tmp_path / "scene.bin", tmp_path / "out.f32")
    assert p.returncode == 0, p.stderr
    got = np.fromfile(tmp_path / "out.f32", dtype=np.float32).reshape(120, 160, 2)
    ref = oracle_port.render(xyz, cells, a, q, rots, 160, 120, mg.REFERENCE_BOUNDS, solid_tets=solid, solid_colour=float("nan"), threads=2)
    assert np.array_equal(got.view(np.uint32), ref["image"].view(np.uint32))
    assert f"segments {ref['segments']} covered {ref['covered']} marked {ref['marked']}" in p.stdout
    # an unrotated lattice-aligned grid: pixel centres on projected edges -> the reference aborts (plane.cpp:39-41),
    # the oracle reports; either way no sanitizer report
    san("oracle", tmp_path / "scene.bin", tmp_path / "out.f32")
    (tmp_path / "short.bin").write_bytes((tmp_path / "scene.bin").read_bytes()[:1000])
    assert san("oracle", tmp_path / "short.bin", tmp_path / "o").returncode == 1


def test_row_blocks_equal_the_python_partition(san):
    """The native host's cost-balanced row blocks (csrc/host/row_blocks.cpp) are the partition course5_amd/sharding.py
    computes for the one-process-per-GPU path: the two multi-GPU paths share a frame the same way."""
    from course5_amd import sharding
    rng = np.random.default_rng(3)
    for trial in range(40):
        n = int(rng.integers(8, 400))
        world = int(rng.integers(1, min(9, n)))
        kind = trial % 4
        cost = (rng.integers(0, 5000, n) if kind == 0 else np.zeros(n, dtype=np.int64) if kind == 1 else
                np.r_[np.zeros(n // 3, dtype=np.int64), rng.integers(1000, 90000, n - 2 * (n // 3)), np.zeros(n // 3, dtype=np.int64)] if kind == 2 else
                np.full(n, 77))
        base = float(rng.choice([0.0, 1.0, 14400.0]))
        if base == 0.0 and cost.sum() == 0:
            base = 1.0
        p = san("blocks", world, base, *[int(c) for c in cost])
        assert p.returncode == 0, p.stderr
        got = [tuple(int(v) for v in line.split()) for line in p.stdout.strip().splitlines()]
        first = sharding.balanced_blocks(cost, world, base_cost=base)
        assert got == first, (trial, n, world, base)
        # (cuts at multiples of the walk's tile height)
        p = san("blocks", world, base, *[int(c) for c in cost], C5_BLOCK_QUANTUM=8)
        assert p.returncode == 0, p.stderr
        got = [tuple(int(v) for v in line.split()) for line in p.stdout.strip().splitlines()]
        want = sharding.balanced_blocks(cost, world, base_cost=base, quantum=8)
        assert got == want, (trial, n, world, base)
        assert sum(k for _, k in want) == n and all(k >= 1 for _, k in want)
        if n >= 16 * world:
            assert all(b % 8 == 0 for b, _ in want), want
        # ... and cut again by the times the devices took for those blocks (time_weighted_row_costs / time_weighted_costs)
        times = [float(np.round(t, 4)) for t in rng.uniform(0.05, 0.5, world)]
        if trial % 5 == 0:
            times[0] = 0.0  # a device without a time keeps its model costs
        p = san("wblocks", world, base, *times, *[int(c) for c in cost])
        assert p.returncode == 0, p.stderr
        got = [tuple(int(v) for v in line.split()) for line in p.stdout.strip().splitlines()]
        assert got == sharding.balanced_blocks(sharding.time_weighted_costs(cost, first, times, base_cost=base), world), (trial, n, world, base, times)
    assert san("blocks", 5, 1.0, 1, 2, 3).returncode == 1  # more devices than rows: a message
