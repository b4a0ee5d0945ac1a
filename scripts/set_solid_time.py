"""Scratch: host time of c5_set_solid (unique points / faces of the tet soups) for the real lobe and sphere."""
import sys, os, subprocess, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from course5_amd import capi, meshgen as mg
d = tempfile.mkdtemp()
xyz, c, a, q = mg.workload("c1")
mg.write_vtk_ascii(f"{d}/c1.vtk", xyz, c, a, q)
t = time.perf_counter()
subprocess.run(["course5_amd/course", "-f", f"{d}/c1.vtk", "-d", f"{d}/o.vti", "--parse_only", "--dump_solids", f"{d}/s.bin"], check=True, capture_output=True)
print(f"course --parse_only --dump_solids (generates lobe + sphere): {1e3 * (time.perf_counter() - t):.0f} ms")
raw = open(f"{d}/s.bin", "rb").read()
off = 0; solids = []
while off < len(raw):
    n = int(np.frombuffer(raw, dtype=np.int64, count=1, offset=off)[0])
    solids.append(np.frombuffer(raw, dtype=np.float64, count=12 * n, offset=off + 8).reshape(n, 12)); off += 8 + 96 * n
ctx = capi.Context(0)
ctx.set_option("view_cache", 0)  # a benchmark of identical frames: each one does its whole per-view setup
for k, name in enumerate(("lobe", "sphere")):
    for rep in range(2):
        t = time.perf_counter(); ctx.set_solid(k, solids[k]); print(f"c5_set_solid {name} ({len(solids[k])} tets): {1e3 * (time.perf_counter() - t):.0f} ms")
