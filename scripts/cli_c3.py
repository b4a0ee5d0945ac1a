"""Scratch: end-to-end `course` run on the C3 grid (ASCII and binary input)."""
import sys, os, subprocess, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from course5_amd import meshgen as mg
d = tempfile.mkdtemp()
xyz, c, a, q = mg.workload("c3")
t = time.time(); mg.write_vtk_binary(f"{d}/c3b.vtk", xyz, c, a, q); print("write binary vtk", round(time.time() - t, 2), "s", os.path.getsize(f"{d}/c3b.vtk") / 1e6, "MB")
for extra in ([], ["--no_solids"], ["--frames", "20", "--sweep", "Y", "--sweep_step", "0.01", "--no_solids"]):
    t = time.time()
    r = subprocess.run(["course5_amd/course", "-f", f"{d}/c3b.vtk", "-d", f"{d}/out.vti", "-x", "2400", "-y", "1800", "-X", "0.1", "-Y", "0.07", "--stats"] + extra, capture_output=True, text=True)
    print(extra, "wall", round(time.time() - t, 2), "s rc", r.returncode)
    print("   ", "\n    ".join(l for l in r.stdout.splitlines() if "completed" in l or "GPU frame" in l or "further" in l), r.stderr[-200:])
