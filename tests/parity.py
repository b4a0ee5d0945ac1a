"""Image comparison used by every parity test.

Bar (BASELINE.json north_star): output pixel values within 1e-5 relative of the
reference render; NaN == NaN (solid pixels).  An absolute floor of 1e-6 * max|image| covers
pixels whose value is ~0 (silhouette chords).  The walk on the GPU is fp64 like the reference, so
in practice images agree bit for bit after the fp32 narrowing; tests report how many values differ
at all.
"""
import glob
import os

import numpy as np

REL_TOL = 1e-5
ABS_FLOOR = 1e-6
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def compare(a: np.ndarray, b: np.ndarray) -> dict:
    """a: under test, b: expected; [rows, cols, 2] float32."""
    assert a.shape == b.shape, (a.shape, b.shape)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    res = dict(nan_mismatch=int((nan_a != nan_b).sum()), outliers=0, differing=0, max_abs=0.0, max_rel=0.0)
    a64 = np.where(nan_a, 0.0, a).astype(np.float64)
    b64 = np.where(nan_b, 0.0, b).astype(np.float64)
    for ch in range(a.shape[-1]):
        A, B = a64[..., ch], b64[..., ch]
        diff = np.abs(A - B)
        tol = REL_TOL * np.maximum(np.abs(A), np.abs(B)) + ABS_FLOOR * (np.abs(B).max() if B.size else 0.0)
        res["outliers"] += int((diff > tol).sum())
        res["differing"] += int((A != B).sum())
        res["max_abs"] = max(res["max_abs"], float(diff.max()) if diff.size else 0.0)
        denom = np.maximum(np.abs(B), 1e-300)
        res["max_rel"] = max(res["max_rel"], float((diff / denom)[np.abs(B) > 1e-3 * np.abs(B).max()].max()) if B.size and np.abs(B).max() > 0 else 0.0)
    return res


def assert_images_match(a, b, what=""):
    r = compare(a, b)
    assert r["nan_mismatch"] == 0, f"{what}: NaN mask differs at {r['nan_mismatch']} values"
    assert r["outliers"] == 0, f"{what}: {r['outliers']} values beyond 1e-5 relative (+1e-6 abs floor): {r}"
    return r


def golden_fixtures():
    return sorted(p for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                  if not p.endswith("rotations.npz") and not os.path.basename(p).startswith("g3_"))


def load_golden(path):
    z = np.load(path)  # allow_pickle=False (default)
    fx = {k: z[k] for k in z.files}
    fx["name"] = os.path.splitext(os.path.basename(path))[0]
    return fx
