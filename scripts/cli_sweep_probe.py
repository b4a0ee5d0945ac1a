import sys, os, subprocess, time, tempfile
sys.path.insert(0, os.getcwd())
from course5_amd import meshgen as mg
d = tempfile.mkdtemp()
xyz, c, a, q = mg.workload("c3")
mg.write_vtk_binary(f"{d}/c3b.vtk", xyz, c, a, q)
for extra in (["--raw_vti"], []):
    for j in ("-j16", "-j4", "-j64"):
        t = time.time()
        r = subprocess.run(["course5_amd/course", "-f", f"{d}/c3b.vtk", "-d", f"{d}/out.vti", "-x", "2400", "-y", "1800", "-X", "0.1", "-Y", "0.07", "--stats", j,
                            "--frames", "10", "--sweep", "Y", "--sweep_step", "0.01", "--no_solids"] + extra, capture_output=True, text=True)
        print(extra, j, "wall", round(time.time() - t, 2), "s;", " | ".join(l.strip() for l in r.stdout.splitlines() if "further" in l or "Ray-tracing" in l))
