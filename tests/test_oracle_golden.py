"""The CPU oracle (oracle/oracle.cpp) against the golden vectors produced by the reference's own
line.cpp / tetra.cpp object code (tests/golden/make_golden.py).  CPU only."""
import hashlib

import numpy as np
import pytest

from parity import golden_fixtures, load_golden, GOLDEN_DIR
import os


@pytest.mark.parametrize("path", golden_fixtures(), ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_reproduces_golden_bit_for_bit(oracle_port, path):
    fx = load_golden(path)
    rx, ry = (int(v) for v in fx["res"])
    stride = int(fx["stride"])
    for k in range(len(fx["views"])):
        r = oracle_port.render(fx["xyz"], fx["cells"], fx["alpha"], fx["q"], fx[f"rots{k}"], rx, ry,
                               fx["bounds"], alpha_limit=float(fx["alpha_limit"]), threads=2)
        img = r["image"]
        assert r["segments"] == int(fx[f"segments{k}"])
        assert r["covered"] == int(fx[f"covered{k}"])
        # integer/bit comparison: the restatement performs the same fp64 operations in the same order
        assert np.array_equal(img[::stride, ::stride].view(np.uint32), fx[f"image{k}"].view(np.uint32))
        assert hashlib.sha256(img.tobytes()).digest() == fx[f"sha256_{k}"].tobytes()


def test_oracle_rotation_known_answers(oracle_port):
    z = np.load(os.path.join(GOLDEN_DIR, "rotations.npz"))
    k = 0
    while f"rots{k}" in z.files:
        got = oracle_port.rotate_points(z["pts"], z[f"rots{k}"])
        assert np.array_equal(got.view(np.uint64), z[f"out{k}"].view(np.uint64))
        k += 1
    assert k == 4
