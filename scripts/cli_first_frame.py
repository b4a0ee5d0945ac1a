"""Scratch: what the `course` CLI reports for ONE frame of the C3 grid, with and without the solids."""
import sys, os, subprocess, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from course5_amd import meshgen as mg
d = tempfile.mkdtemp()
xyz, c, a, q = mg.workload("c3")
mg.write_vtk_binary(f"{d}/c3b.vtk", xyz, c, a, q)
for extra in ([], ["--no_solids"]):
    for rep in range(2):
        t = time.time()
        r = subprocess.run(["course5_amd/course", "-f", f"{d}/c3b.vtk", "-d", f"{d}/out.vti", "-x", "2400", "-y", "1800", "-X", "0.1", "-Y", "0.07", "-j16"] + extra,
                           capture_output=True, text=True)
        print(extra, "wall", round(time.time() - t, 2), "s;", " | ".join(l.strip() for l in r.stdout.splitlines() if "completed in" in l))
