"""Scratch: one fuzz scene in detail.  python scripts/fuzz_debug.py SEED"""
import os, sys
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util
spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tests", "fuzz_parity.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
from course5_amd import capi, meshgen as mg
from oracle.pyoracle import Oracle
seed = int(sys.argv[1])
xyz, cells, alpha, q, rots, res, limit = fz.scene(seed)
o = Oracle("port")
ref = o.render(xyz, cells, alpha, q, rots, res[0], res[1], mg.REFERENCE_BOUNDS, alpha_limit=limit, threads=8)
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
ctx.upload_grid(xyz, cells, alpha, q); ctx.set_image(res[0], res[1], mg.REFERENCE_BOUNDS); ctx.set_view(rots); ctx.set_alpha_limit(limit)
print("scene", seed, "cells", len(cells), "res", res, "S ref", ref["segments"], "covered", ref["covered"])
for order in (0, 1):
    for key in (1, 0):
        ctx.set_option("integration", order); ctx.set_option("entry_key", key)
        img = ctx.render(); st = ctx.stats()
        a, b = img.astype(np.float64), ref["image"].astype(np.float64)
        tol = 1e-5 * np.maximum(np.abs(a), np.abs(b)) + 1e-6 * np.abs(b).max()
        bad = np.argwhere((np.abs(a - b) > tol).any(axis=-1))
        diff = np.argwhere((img.view(np.uint32) != ref["image"].view(np.uint32)).any(axis=-1))
        print(f"order {order} entry_key {key}: S {st['segments']} (ref {ref['segments']}), entries {st['entries']}, beyond tol {len(bad)}, differing at all {len(diff)}")
        show = bad if len(bad) else diff
        for (r, c) in show[:6]:
            pr = o.render(xyz, cells, alpha, q, rots, res[0], res[1], mg.REFERENCE_BOUNDS, alpha_limit=limit, threads=2, probes=[(c, r)])["probes"][0]
            print(f"   pixel row {r} col {c}: gpu {img[r, c]} ref {ref['image'][r, c]}; oracle segments {len(pr)}: "
                  + " ".join(f"(t{int(t)} z{z:.9f} dz{dz:.3e})" for t, z, dz in pr[:40]))
