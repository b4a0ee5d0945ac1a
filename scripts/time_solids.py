"""Scratch: cost of the solid mask raster with the real lobe + sphere at 2400x1800."""
import sys, os, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from course5_amd import capi, meshgen as mg
d = tempfile.mkdtemp()
xyz, c, a, q = mg.workload("c1")
mg.write_vtk_ascii(f"{d}/c1.vtk", xyz, c, a, q)
subprocess.run(["course5_amd/course", "-f", f"{d}/c1.vtk", "-d", f"{d}/o.vti", "--parse_only", "--dump_solids", f"{d}/s.bin"], check=True, capture_output=True)
raw = open(f"{d}/s.bin", "rb").read()
off = 0; solids = []
while off < len(raw):
    n = int(np.frombuffer(raw, dtype=np.int64, count=1, offset=off)[0])
    solids.append(np.frombuffer(raw, dtype=np.float64, count=12 * n, offset=off + 8).reshape(n, 12)); off += 8 + 96 * n
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
xyz, c, a, q = mg.workload("c3")
ctx.upload_grid(xyz, c, a, q)
ctx.set_image(2400, 1800, mg.REFERENCE_BOUNDS)
rots = mg.view_rotations(0.1, 0.07)
ctx.set_view(rots)
ctx.set_option("solid_cache", 0)
for interior in (0, 1, 0, 1):
  ctx.set_option("solid_interior_faces", interior)
  for which in ("lobe", "sphere", "both"):
    ctx.set_solid(0, solids[0] if which != "sphere" else np.zeros((0, 12))); ctx.set_solid_view(0, np.vstack([[1.0, 0.0, 1.0], rots]))
    ctx.set_solid(1, solids[1] if which != "lobe" else np.zeros((0, 12))); ctx.set_solid_view(1, np.zeros((0, 3)))
    for i in range(6):
        img = ctx.render(); st = ctx.stats()
    print("interior faces rastered" if interior else "interior faces skipped ", which, "ms_solids", round(st["ms_solids"], 3), "solid px", st["solid_pixels"], "total", round(st["ms_total"], 3), flush=True)
