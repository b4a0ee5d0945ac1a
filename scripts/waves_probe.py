"""A/B of walk builds in separate processes is unreliable (clocks); this runs ONE library per process but
interleaves nothing — use only for coarse differences (> 3 %).  C5_LIB selects the build."""
import sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from course5_amd import capi, meshgen as mg  # noqa: E402
xyz, cells, alpha, q = mg.workload("c3")
ctx = capi.Context(0)
ctx.upload_grid(xyz, cells, alpha, q)
ctx.set_image(2400, 1800, mg.REFERENCE_BOUNDS)
ctx.set_view(mg.view_rotations(**mg.BENCH_VIEW))
ctx.set_option("stage_timing", 0)
out = torch.zeros((1800, 2400, 2), dtype=torch.float32, device="cuda:0")
def run(n):
    for _ in range(n):
        ctx.render_device(out.data_ptr())
    ctx.synchronize()
    return ctx.walk_kernel_ms(reset=True)[0]
for prec, rays in ((0, 1), (1, 1), (1, 2), (1, 1), (1, 2)):
    ctx.set_option("precision", prec)
    ctx.set_option("rays_per_lane", rays)
    run(200)
    print(capi.LIB_PATH.split("/")[-1], "precision", prec, "rays per lane", rays, "walk %.4f ms" % run(400), flush=True)
