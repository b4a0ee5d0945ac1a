import sys, os, subprocess, tempfile
sys.path.insert(0, os.getcwd())
from course5_amd import meshgen as mg
d = tempfile.mkdtemp()
xyz, c, a, q = mg.workload("c3")
mg.write_vtk_binary(f"{d}/c3b.vtk", xyz, c, a, q)
r = subprocess.run(["course5_amd/course", "-f", f"{d}/c3b.vtk", "-d", f"{d}/out.vti", "-x", "2400", "-y", "1800", "-X", "0.1", "-Y", "0.07", "-j16",
                    "--frames", "40", "--sweep", "Y", "--sweep_step", "0.00555556", "--no_solids"], capture_output=True, text=True, env=dict(os.environ, C5_VTI_TIMING="1"))
lines = [l for l in r.stderr.splitlines() if l.startswith("write_vti")]
import re
vals = [[float(x) for x in re.findall(r"([0-9.]+) (?:pack|base64|write|ms)", l)] for l in lines]
print(len(lines), lines[-3:])
print([l for l in r.stdout.splitlines() if "further" in l or "Ray-tracing" in l])
print("file bytes", os.path.getsize(f"{d}/out_0001.vti") if os.path.exists(f"{d}/out_0001.vti") else [f for f in os.listdir(d)][:5])
