import os
import sys

import pytest

# torch ships its own HIP runtime; it must be loaded before libcourse5_hip.so pulls in the system
# one, or torch later finds no GPU in this process (tests that hand torch tensors to the C ABI).
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_sessionstart(session):
    """A fresh checkout has no binaries (they are git-ignored): build them once (hipcc cross-compiles
    gfx950 without a GPU).  On the GPU box the prebuilt files travel with the snapshot."""
    import shutil
    from course5_amd import build as c5build
    if not (os.path.exists(c5build.LIB) and os.path.exists(c5build.CLI)) and shutil.which("hipcc"):
        c5build.build_all()


@pytest.fixture(scope="session")
def oracle_port():
    """CPU restatement of the reference algorithm (oracle/oracle.cpp) — the checker."""
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle.Oracle("port")


@pytest.fixture(scope="session")
def oracle_ref():
    """Reference-backed checker (real line.cpp / tetra.cpp); only where oracle/_ref was built."""
    from oracle import pyoracle
    if not pyoracle.reference_available():
        pytest.skip("oracle/_ref/libcourse5_ref.so not built (needs /root/reference)")
    return pyoracle.Oracle("reference")


@pytest.fixture(scope="session")
def gpu_ctx():
    """One c5_context on cuda:0 through the C ABI.  Fails loudly without the HIP library/GPU."""
    from course5_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def product_solids(tmp_path_factory):
    """The Roche lobe and the accretor sphere as the `course` CLI generates them (csrc/host/scene.cpp,
    restating object3d_base.cpp:83-196): raw soups [n][4][3], before any rotation.  Host-only."""
    import subprocess
    import numpy as np
    from course5_amd import meshgen as mg
    d = tmp_path_factory.mktemp("solids")
    xs, cs = mg.cube8()
    mg.write_vtk_ascii(str(d / "tiny.vtk"), xs, cs, *mg.scalars(len(cs)))
    subprocess.run([os.path.join(ROOT, "course5_amd", "course"), "-f", str(d / "tiny.vtk"), "-d", str(d / "o.vti"),
                    "--parse_only", "--dump_solids", str(d / "s.bin")], check=True, capture_output=True)
    raw = open(d / "s.bin", "rb").read()
    off, soups = 0, []
    while off < len(raw):
        n = int(np.frombuffer(raw, dtype=np.int64, count=1, offset=off)[0])
        soups.append(np.frombuffer(raw, dtype=np.float64, count=12 * n, offset=off + 8).reshape(n, 4, 3).copy())
        off += 8 + 96 * n
    assert [len(s) for s in soups] == [130_560, 522_242]
    return soups
