#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the render hot path at 2400x1800 on the ~1M-tet synthetic grid.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one frame through the whole hot path on data already resident in HBM: view transform ->
per-cell records -> boundary entry raster -> walk_composite -> fp32 image in device memory
(N > 1: each rank renders its rows, then one RCCL exchange per frame brings the strips to rank 0).
Rank 0 prints ONE JSON line.

value       whole-job Mrays/s = res_x * res_y * K / (max over ranks of the timed region) / 1e6
roofline    of the dominant kernel walk_composite, every fraction <= 1 by construction:
              achieved / peak / frac   HBM-side bytes per launch (rocprofv3 PMC, profiles/<round>_roofline.json:
                                       (2 x FETCH_SIZE + WRITE_SIZE) KiB, MI355X_MICROARCH.md HBM section) /
                                       the kernel's average duration measured LIVE here with HIP events on the
                                       stream it runs on, against the 8 TB/s HBM3E peak;
              limiter                  what actually binds the kernel: the busiest unit by the same PMC passes
                                       (VALU pipes, LDS, L2 requests, wavefront slots);
              contract                 SURVEY.md section 8(d)'s algorithmic figure (S * 144 B + P * 8 B) / duration —
                                       NOT a fraction of anything: the 128 MB of per-view records are re-read
                                       from L2 / Infinity Cache ~130 times per ray, so it exceeds the HBM peak.
cpu_baseline  the CPU oracle (own restatement of the reference algorithm, OpenMP) timed on this box's
            host cores on the SAME workload at the SAME image size (rank 0, N = 1 only).  It is the checker
            being timed as a baseline, never the product path.
"""
from __future__ import annotations

import argparse
import json
import os
import resource
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from course5_amd import capi, meshgen as mg, sharding  # noqa: E402
from course5_amd.pipeline import FramePipeline, gather_row_costs  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
HBM_ACHIEVABLE_GBS = 6290.0  # measured float4 copy, same table
L2_PEAK_GBS = 34500.0        # aggregate L2, MI355X_MICROARCH.md "L2 (per XCD)"
B_SEG_SURVEY = 144           # 16 cell->vertex + 16 adjacency + 96 vertices + 16 scalars
B_SEG_RECORD = 128           # what walk_composite actually reads per step: one 128 B ExitRecord (exit planes, neighbours, optics)
B_PIX = 8                    # 2 x fp32 store per pixel
TILE_ROWS = 16
ROOFLINE_FILE = os.path.join(ROOT, "profiles", "roofline.json")  # written by scripts/summarize_profile.py


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=1000)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--workload", default="c3", help="c3 (998 250 tets, default), c2, kuhnN")
    p.add_argument("--res", default="2400x1800", help="image size of the N = 1 workload")
    p.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                   help="N > 1: strong (default) = the same --res frame split over N GPUs: north_star's \">= 6x at 8 GPUs on "
                        "row-tile split\" clause, and the workload of the N = 1 line, so that value(N) / value(1) is a "
                        "speed-up; weak = per-GPU rays stay at --res (the image grows by sqrt(N) per side)")
    p.add_argument("--no-configs", action="store_true",
                   help="N > 1: skip the figures of BASELINE configs 4 (4800x3600 split by rows) and 5 (360-frame -D sweep "
                        "with solids, whole frames dealt to the GPUs) that follow the headline")
    p.add_argument("--config4-res", default="4800x3600", help="image size of the config-4 figure")
    p.add_argument("--config5-frames", type=int, default=360, help="frames of the config-5 sweep (D = k / 180)")
    p.add_argument("--mode", choices=["rows", "frames"], default="rows",
                   help="N > 1: rows = every frame is split by rows over the ranks and exchanged (config 4); "
                        "frames = frame k of a sweep is rendered whole by rank k mod N, no exchange (config 5)")
    p.add_argument("--tile", type=int, default=-1, help="wavefront tile shape override (0: 64x1, 1: 16x4, 2: 8x8 in workgroups of four, 3: 8x8 one per workgroup)")
    p.add_argument("--lds-stage", type=int, default=-1, help="override: 2 = LDS-DMA staging (default), 1 = staged through vector registers, 0 = direct loads")
    p.add_argument("--pipeline", type=int, default=-1, help="override: overlap the next frame's setup with the walk (1) or not (0)")
    p.add_argument("--overlap-setup", type=int, default=-1, help="override: entry lists beside build_records (1) or serial (0)")
    p.add_argument("--own-stream", action="store_true", help="run on the context's own (high priority) stream")
    p.add_argument("--sweep", choices=["none", "Y", "D"], default="none",
                   help="advance -Y (view) or -D (donor angle, needs --solids) by 1/180 pi per step: BASELINE config 5")
    p.add_argument("--solids", action="store_true", help="add the Roche lobe and the accretor sphere (generated by the course CLI)")
    p.add_argument("--backend", default="nccl", help="nccl (= RCCL, default); gloo only to rehearse N > 1 on one GPU")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-cpu-reference", action="store_true", help="skip the second CPU figure: the reference's own line.cpp / tetra.cpp object code, one thread (about half a minute)")
    p.add_argument("--no-host-image", action="store_true", help="skip the second figure: frames delivered to host memory")
    p.add_argument("--no-native", action="store_true", help="skip the figures of the C++ host (`course --bench`)")
    p.add_argument("--no-steady", action="store_true", help="skip the clock-steadying frames after the W warm-up steps (profiling passes)")
    p.add_argument("--cpu-sample-res", default="", help="image size of the CPU baseline (default: the benchmark's own)")
    p.add_argument("--pipeline-depth", type=int, default=2, help="N > 1: frames whose exchange may be in flight")
    p.add_argument("--sharding", choices=["blocks", "cyclic"], default="blocks",
                   help="N > 1: contiguous cost-balanced row blocks (default: every block is received at its final offset "
                        "of rank 0's frame, no reassembly pass; every rank builds only the records its rays can reach) or "
                        "cyclic 16-row tiles (one gather of padded strips + a reassembly pass on rank 0)")
    p.add_argument("--row-base-cost", type=float, default=3.0,
                   help="blocks: fixed cost per pixel in segment units added when balancing (measured: an empty "
                        "pixel costs about six segments)")
    return p.parse_args()


def usable_cpus() -> int:
    """CPUs this process may really use: the affinity mask, cut down to a cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(xyz, cells, alpha, q, rots, res):
    """Oracle ("port") on the host cores: same grid, same view, same image size."""
    from oracle.pyoracle import Oracle
    cores = min(usable_cpus(), 32)  # 32 = MAX_NUMBER_OF_THREADS (config.hpp:39)
    rx, ry = res
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    r = Oracle("port").render(xyz, cells, alpha, q, rots, rx, ry, mg.REFERENCE_BOUNDS, threads=cores)
    rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    ctor, binning, resolve = (float(v) for v in r["timing_ms"])
    span = ctor + binning + resolve
    return {
        "value": rx * ry / span / 1e3, "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": f"one whole frame of the same workload: same grid, view and image size {rx}x{ry} "
                  f"({r['segments']} segments) on {cores} OpenMP threads; span = pixel grid + binning + resolve "
                  f"(the reference's own timed span, main.cpp:126-130) = {span:.0f} ms; binning + resolve only = "
                  f"{binning + resolve:.0f} ms; peak RSS of the process {rss1 / 1e6:.2f} GB (before: {rss0 / 1e6:.2f})",
        "value_bin_resolve_only": rx * ry / (binning + resolve) / 1e3,
    }, r["segments"]


def cpu_baseline_reference(xyz, cells, alpha, q, rots, res):
    """The reference-backed checker (oracle/_ref: the reference's OWN line.cpp + tetra.cpp object code, compiled in
    the build container by oracle/Makefile, driven serially by oracle/ref_driver.cpp; plane.cpp's scan conversion is
    the restatement of oracle/scan.hpp because plane.cpp needs VTK headers) on the same grid, view and image size:
    one frame on ONE core, about half a minute.  None where the prebuilt library did not travel."""
    from oracle import pyoracle
    if not pyoracle.reference_available():
        return None
    rx, ry = res
    t0 = time.perf_counter()
    r = pyoracle.Oracle("reference").render(xyz, cells, alpha, q, rots, rx, ry, mg.REFERENCE_BOUNDS)
    dt = time.perf_counter() - t0
    return {"value": rx * ry / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "reference",
            "sample": f"one whole frame of the same workload ({rx}x{ry}, {r['segments']} segments) through the reference's own line.cpp / "
                      f"tetra.cpp object code (808-byte line objects, per-pixel std::sort, libm exp), one thread: {dt:.1f} s from the "
                      "rotation of the tets to the last pixel; plane.cpp's binning restated (oracle/scan.hpp), no VTK load; "
                      "BASELINE.md section 2 has the complete reference binary at 0.13-0.34 Mrays/s on 8 cores"}


def load_roofline_counters():
    """Per-launch PMC figures of walk_composite from the committed rocprofv3 summary, if any."""
    try:
        with open(ROOFLINE_FILE) as f:
            return json.load(f)
    except Exception:
        return None


def predicted_scaling(res_x, res_y, world):
    """What row-splitting this frame over `world` GPUs can give at best: every rank's share of the frame timed in turn
    on ONE GPU (scripts/sim_scaling.py -> profiles/sim_scaling.json; no exchange, no second GPU involved)."""
    try:
        with open(os.path.join(ROOT, "profiles", "sim_scaling.json")) as f:
            sim = json.load(f)
        e = sim["frames"][f"{res_x}x{res_y}"]["world"][str(world)]
        return {"speedup": e["blocks"]["speedup"], "slowest_rank_ms": e["blocks"]["max_ms"], "one_gpu_ms": sim["frames"][f"{res_x}x{res_y}"]["full_ms"],
                "cyclic_speedup": e["cyclic"]["speedup"],
                "what": "per-rank GPU times of the balanced blocks measured one after the other on one GPU "
                        f"({sim.get('round', '?')}, scripts/sim_scaling.py): a ray is a chain of ~130 dependent steps, so a "
                        "rank's time does not fall in proportion to its rows; exchange not included"}
    except Exception:  # noqa: BLE001
        return None


def product_solids():
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        xs, cs = mg.cube8()
        mg.write_vtk_ascii(f"{d}/tiny.vtk", xs, cs, *mg.scalars(len(cs)))
        subprocess.run([os.path.join(ROOT, "course5_amd", "course"), "-f", f"{d}/tiny.vtk", "-d", f"{d}/o.vti",
                        "--parse_only", "--dump_solids", f"{d}/s.bin"], check=True, capture_output=True)
        raw = open(f"{d}/s.bin", "rb").read()
    off, soups = 0, []
    while off < len(raw):
        n = int(np.frombuffer(raw, dtype=np.int64, count=1, offset=off)[0])
        soups.append(np.frombuffer(raw, dtype=np.float64, count=12 * n, offset=off + 8).reshape(n, 12))
        off += 8 + 96 * n
    return soups


def native_course_bench(xyz, cells, alpha, q, res_x, res_y, n_devices, frames, warmup, variants, rounds=5):
    """The C++ host (`course`, one process driving every GPU through one c5_context each) on the same grid,
    view and image: frames rendered and delivered to pinned host memory, no files.  Secondary figures; a
    failure here is recorded, never fatal."""
    import subprocess
    import tempfile
    out = {}
    exe = os.path.join(ROOT, "course5_amd", "course")
    try:
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "grid.vtk")
            mg.write_vtk_binary(src, xyz, cells, alpha, q, v51=True)
            devs = ",".join(str(k) for k in range(n_devices)) if not os.environ.get("C5_BENCH_ONE_DEVICE") else ",".join(["0"] * n_devices)
            for name, extra in variants:
                cmd = [exe, "-f", src, "-d", os.path.join(d, "frame.vti"), "--no_solids", "-x", str(res_x), "-y", str(res_y), "-X", str(mg.BENCH_VIEW["angle_around_x"]),
                       "-Y", str(mg.BENCH_VIEW["angle_around_y"]), "--bench", str(frames), "--bench_warmup", str(warmup),
                       "--bench_rounds", str(rounds), "--sweep", "Y", "--sweep_step", "0", "--devices", devs] + extra
                try:
                    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{"course_bench"')]
                    if r.returncode == 0 and line:
                        out[name] = json.loads(line[0])["course_bench"]
                        if "instead" in r.stderr:
                            out[name]["note"] = [ln for ln in r.stderr.splitlines() if "instead" in ln][0]
                    else:
                        out[name] = {"error": (r.stderr or r.stdout)[-300:]}
                except Exception as e:  # noqa: BLE001
                    out[name] = {"error": repr(e)[:300]}
    except Exception as e:  # noqa: BLE001
        out["error"] = repr(e)[:300]
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: start the N ranks as a CHILD job (one process per GPU under
        # torch.distributed.run, exactly what the driver's documented command line does) BEFORE anything here has
        # touched the GPU, hand its output through and leave with its exit code.  Never an exec.
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if os.environ.get("C5_BENCH_ONE_DEVICE"):  # rehearsal only: every rank on GPU 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rdev = dev if args.backend == "nccl" else torch.device("cpu")  # where small reductions live
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        # host-side barriers for the phases in which the GPUs must be left alone (an RCCL barrier is a kernel that
        # spins on every GPU until the last rank arrives)
        cpu_group = dist.new_group(backend="gloo") if args.backend == "nccl" else None
        # who is here: every rank's device ordinal, as the communicator itself sees the job (a SCALE record proves N ranks)
        seen = [None] * world
        dist.all_gather_object(seen, {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(),
                                      "name": torch.cuda.get_device_name(dev), "pid": os.getpid()}, group=cpu_group)

    res_x, res_y = (int(v) for v in args.res.lower().split("x"))
    base_res = (res_x, res_y)
    frame_parallel = world > 1 and args.mode == "frames"
    if world > 1 and args.scaling == "weak" and not frame_parallel:
        res_x, res_y = int(round(res_x * world ** 0.5)), int(round(res_y * world ** 0.5))
    xyz, cells, alpha, q = mg.workload(args.workload)
    rots = mg.view_rotations(**mg.BENCH_VIEW)

    stream = torch.cuda.Stream(device=dev)
    ctx = capi.Context(local_rank)
    if args.pipeline >= 0:
        ctx.set_option("pipeline", args.pipeline)
    ctx.upload_grid(xyz, cells, alpha, q)
    ctx.set_view(rots)
    ctx.set_alpha_limit(2.5)
    ctx.set_option("stage_timing", 0)
    if args.tile >= 0:
        ctx.set_option("tile", args.tile)
    if args.lds_stage >= 0:
        ctx.set_option("lds_stage", args.lds_stage)
    if args.overlap_setup >= 0:
        ctx.set_option("overlap_setup", args.overlap_setup)
    if not args.own_stream:
        ctx.set_stream(stream.cuda_stream)

    soups = None

    def put_solids(on: bool):
        nonlocal soups
        if on:
            if soups is None:
                soups = product_solids()
            ctx.set_solid(0, soups[0])
            ctx.set_solid(1, soups[1])
            ctx.set_solid_view(1, np.zeros((0, 3)))  # the sphere is never rotated (main.cpp:116)
            ctx.set_solid_view(0, np.vstack([[1.0, 0.0, 1.0], rots]))
        else:
            ctx.set_solid(0, np.zeros((0, 12)))
            ctx.set_solid(1, np.zeros((0, 12)))

    if args.solids:
        put_solids(True)

    # "view_cache" (a frame with the view of the frames before it reuses their per-view data: the persistent grid of a
    # donor sweep): OFF wherever this benchmark renders one view over and over only because it is a benchmark - every
    # timed frame of the headline does the whole per-view setup -, ON (the product's default) for the -D sweep, whose
    # view is fixed by definition (main.cpp:112-116: only the donor turns)
    ctx.set_option("view_cache", 1 if args.sweep == "D" else 0)
    # frame k of a sweep: in frame-parallel mode rank r renders frames r, r + N, r + 2N, ...
    sweep = {"kind": args.sweep, "k": rank if frame_parallel else 0, "stride": world if frame_parallel else 1, "solids": bool(args.solids)}

    def advance_view():
        if sweep["kind"] == "none":
            return
        ang = sweep["k"] / 180.0
        sweep["k"] += sweep["stride"]
        if sweep["kind"] == "Y":
            r = mg.view_rotations(mg.BENCH_VIEW["angle_around_x"], mg.BENCH_VIEW["angle_around_y"] + ang)
            ctx.set_view(r)
            if sweep["solids"]:
                ctx.set_solid_view(0, np.vstack([[1.0, 0.0, 1.0], r]))
        elif sweep["solids"]:
            ctx.set_solid_view(0, np.vstack([[1.0, ang * 3.14159265358979323846, 1.0], rots]))

    def render(strip):
        advance_view()
        ctx.render_device(strip.data_ptr())

    def host_barrier():
        """The ranks meet without touching the GPUs (an RCCL barrier is a kernel spinning on every GPU)."""
        if world > 1:
            dist.barrier(group=cpu_group) if cpu_group is not None else dist.barrier()

    def lay_out_rows(rx, ry, split):
        """Image size + which rows this rank renders.  Returns the blocks (None: cyclic tiles or the whole image)."""
        ctx.set_row_tiles(0, 0, 1)
        ctx.set_row_range(0, -1)
        ctx.set_image(rx, ry, mg.REFERENCE_BOUNDS)
        if not split:
            return None
        if args.sharding == "cyclic":
            ctx.set_row_tiles(TILE_ROWS, rank, world)
            return None
        # one probe frame on equal blocks measures segments per row; every rank then derives the
        # same cost-balanced contiguous blocks (pixels are independent: plane.cpp:161-169)
        eq = sharding.equal_blocks(ry, world)
        ctx.set_row_range(*eq[rank])
        ctx.set_option("row_costs", 1)
        probe = torch.zeros((eq[rank][1], rx, 2), dtype=torch.float32, device=dev)
        for _ in range(4):
            render(probe)
            if ctx.synchronize() == capi.C5_OK:
                break
        costs = gather_row_costs(ctx.row_costs(), eq, rank, world, rdev)
        blocks = sharding.balanced_blocks(costs, world, base_cost=rx * args.row_base_cost, quantum=8)
        ctx.set_option("row_costs", 0)
        ctx.set_row_range(*blocks[rank])
        del probe
        # ... and cut again, twice, by what the ranks TOOK for those blocks (GPU time of a frame, HIP events): the model
        # misses what a segment costs where and what a share costs whatever its rows (sharding.time_weighted_costs;
        # `course --devices` does the same with the times of its probe frames)
        ctx.set_option("stage_timing", 1)  # (the frame's GPU time comes from its stage events; off again below)
        for _ in range(2):
            buf = torch.zeros((ctx.local_rows, rx, 2), dtype=torch.float32, device=dev)
            mine = 0.0
            for k in range(12):
                render(buf)
                if ctx.synchronize() == capi.C5_OK and k >= 6:
                    t = ctx.stats()["ms_total"]
                    mine = t if mine == 0.0 else min(mine, t)
            del buf
            times = torch.zeros(world, dtype=torch.float64, device=rdev)
            times[rank] = mine
            dist.all_reduce(times)
            times = [float(v) for v in times.cpu()]
            again = sharding.balanced_blocks(sharding.time_weighted_costs(costs, blocks, times, base_cost=rx * args.row_base_cost), world, quantum=8)
            if again == blocks or min(times) <= 0.0:
                break
            blocks = again
            ctx.set_row_range(*blocks[rank])
        ctx.set_option("stage_timing", 0)
        return blocks

    def measure(rx, ry, steps, warmup, split, steadying):
        """W warm-up frames (+ clock-steadying batches), then exactly `steps` frames between barriers +
        torch.cuda.synchronize() on both sides; a C5_RETRY inside the region re-times it."""
        with torch.cuda.stream(stream):
            blocks = lay_out_rows(rx, ry, split)
            n_local = ctx.local_rows
            # rows split: no strip is exchanged before its render is known to be complete (FramePipeline.check): a
            # C5_RETRY is settled by the rank it happened on, before its one exchange of the step, so the ranks
            # never disagree on the number of collectives.  Whole frames: frames run back to back and the status
            # is read once after the timed region (a retry there re-times the region and is reported).
            pipe = FramePipeline(rx, ry, rank, world if split else 1, dev, depth=args.pipeline_depth, tile_rows=TILE_ROWS,
                                 blocks=blocks, host_staging=args.backend != "nccl", check=ctx.synchronize if split else None)
            retries = 0

            def settle():
                """After a burst without per-frame checks: wait, and render again while the library says C5_RETRY."""
                nonlocal retries
                pipe.drain()
                for _ in range(4):
                    if ctx.synchronize() == capi.C5_OK:
                        return
                    retries += 1
                    pipe.step(render)
                    pipe.drain()
                raise SystemExit("frames kept being reported incomplete (C5_RETRY)")

            k0 = sweep["k"]
            for k in range(max(warmup, 1)):
                pipe.step(render)
                if k < 2 and not split:
                    settle()  # the first frames size the internal buffers
            settle()
            # clock-steadying frames (untimed, not counted in W): a 20-frame warm-up is 15 ms, far too short for
            # the clocks to settle (round 1: 0.670 ms per walk in the driver's short run vs 0.606 sustained).
            # Batches of 50 frames until the walk time of a batch is within 1 % of the one before (at most 2 s).
            steady = []
            if steadying:
                ctx.walk_kernel_ms(reset=True)
                t_lim = time.perf_counter() + 2.0
                while time.perf_counter() < t_lim:
                    for _ in range(50):
                        pipe.step(render)
                    pipe.drain()
                    ms, _ = ctx.walk_kernel_ms(reset=True)
                    steady.append(ms)
                    go_on = torch.tensor([0 if (len(steady) >= 2 and abs(steady[-1] - steady[-2]) <= 0.01 * steady[-1]) else 1],
                                         dtype=torch.int64, device=rdev)
                    if world > 1:  # every rank runs the same number of batches (they exchange in every step)
                        dist.all_reduce(go_on)
                    if int(go_on.item()) == 0:
                        break
                settle()
            sweep["k"] = k0
            stats = ctx.stats()
            ctx.walk_kernel_ms(reset=True)
            # (the timed region: the walk kernel is timed by HIP events around every 4th launch - two events cost 6 us of a
            # 0.53-ms frame; the JSON says how many launches the average is taken over)
            walk_every = 4 if steps >= 200 else 1  # (a short region: every launch)
            ctx.set_option("walk_timing", walk_every)
            for attempt in range(3):
                if world > 1:
                    dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(steps):
                    pipe.step(render)
                pipe.drain()
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                elapsed = time.perf_counter() - t0
                # A frame that made an internal buffer grow (C5_RETRY) inside a region without per-frame checks
                # invalidates that region on every rank: time the K steps again.  A hard error is an error.
                rc = ctx.synchronize()
                redo = torch.tensor([1 if rc == capi.C5_RETRY else 0], dtype=torch.int64, device=rdev)
                if world > 1:
                    dist.all_reduce(redo)
                walk_ms, walk_launches = ctx.walk_kernel_ms(reset=True)
                if int(redo.item()) == 0:
                    break
                retries += 1
                sweep["k"] = k0
            else:
                raise SystemExit("frames kept being re-rendered inside the timed region")
            ctx.set_option("walk_timing", 1)
            retries += pipe.retries
        el = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        seg = torch.tensor([stats["segments"], n_local * rx, retries], dtype=torch.int64, device=rdev)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            dist.all_reduce(seg, op=dist.ReduceOp.SUM)
        S_total, P_total, retries_total = (int(v) for v in seg.tolist())
        return {"elapsed": float(el.item()), "steps": steps, "blocks": blocks, "n_local": n_local, "stats": stats,
                "walk_ms": walk_ms, "walk_launches": walk_launches, "steady": steady, "S_total": S_total, "P_total": P_total,
                "retries": retries_total}

    def one_gpu_reference(rx, ry, frames):
        """Rank 0 renders the WHOLE rx x ry frame alone, the other ranks wait on the host: the N = 1 time of the very
        image the N ranks just shared, in the same process, clocks and run."""
        host_barrier()
        ms = None
        if rank == 0:
            with torch.cuda.stream(stream):
                lay_out_rows(rx, ry, False)
                whole = torch.zeros((ry, rx, 2), dtype=torch.float32, device=dev)
                for _ in range(3):
                    for _ in range(20):
                        render(whole)
                    if ctx.synchronize() == capi.C5_OK:
                        break
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(frames):
                        render(whole)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    if ctx.synchronize() == capi.C5_OK:
                        break
                ms = dt * 1e3 / frames
                del whole
        host_barrier()
        return ms

    sharded = world > 1 and not frame_parallel
    head = measure(res_x, res_y, args.steps, args.warmup, sharded, not args.no_steady)
    elapsed, stats, blocks, n_local = head["elapsed"], head["stats"], head["blocks"], head["n_local"]
    walk_ms, walk_launches, steady, retries_total = head["walk_ms"], head["walk_launches"], head["steady"], head["retries"]
    S_total, P_total = head["S_total"], head["P_total"]
    k_ref = max(10, min(args.steps, 200))
    head_one_gpu_ms = one_gpu_reference(res_x, res_y, k_ref) if sharded else None

    host_image = None
    if world == 1 and not args.no_host_image and hasattr(ctx, "render_host_async"):
        host_image = ctx.bench_host_frames(max(100, min(args.steps, 400)))

    # N > 1: BASELINE's other multi-GPU configurations, each beside the one-GPU time of the same work measured by rank 0
    # in this run.  Config 4: ONE 4800x3600 frame of the C3 grid split by rows.  Config 5: the 360-frame -D sweep with
    # the Roche lobe and the accretor sphere resident, whole frames dealt to the GPUs (frame k on rank k mod N, no
    # exchange: utility/rotate_traces.py runs one process per frame).
    config4 = config5 = None
    if sharded and not args.no_configs and args.sweep == "none" and not args.solids:
        c4x, c4y = (int(v) for v in args.config4_res.lower().split("x"))
        k4 = max(10, min(args.steps, 100))
        m4 = measure(c4x, c4y, k4, min(args.warmup, 10), True, False)
        one4 = one_gpu_reference(c4x, c4y, max(5, k4 // 2))
        ms4 = m4["elapsed"] * 1e3 / k4
        config4 = {"what": f"BASELINE config 4: one {c4x}x{c4y} frame of the same grid split by rows over {world} GPUs, one exchange "
                           "per frame to rank 0; STRONG scaling against the one-GPU time of the same image (rank 0 alone, same run)",
                   "value": round(c4x * c4y * k4 / m4["elapsed"] / 1e6, 2), "unit": "Mrays/s", "ms_per_frame": round(ms4, 4),
                   "frames": k4, "rows_per_rank": [n for _, n in m4["blocks"]] if m4["blocks"] else f"cyclic tiles of {TILE_ROWS} rows",
                   "segments_per_frame": m4["S_total"], "retries": m4["retries"]}
        if rank == 0 and one4:
            config4.update(one_gpu_ms_per_frame=round(one4, 4), one_gpu_value=round(c4x * c4y / one4 / 1e3, 2),
                           speedup_vs_one_gpu=round(one4 / ms4, 3))
        # config 5
        n5 = max(world, args.config5_frames)
        per_rank = (n5 + world - 1) // world
        with torch.cuda.stream(stream):
            lay_out_rows(base_res[0], base_res[1], False)
            put_solids(True)
            whole = torch.zeros((base_res[1], base_res[0], 2), dtype=torch.float32, device=dev)
            sweep.update(kind="D", solids=True, stride=world)
            ctx.set_option("view_cache", 1)  # config 5 is the donor sweep: the view is fixed, the grid persistent

            def sweep_frames(first, count, stride):
                sweep["k"], sweep["stride"] = first, stride
                for _ in range(count):
                    render(whole)

            for _ in range(3):  # sizes the buffers, fills the sphere's cached mask
                sweep_frames(rank, 6, world)
                if ctx.synchronize() == capi.C5_OK:
                    break
            for attempt in range(3):
                dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                mine = len(range(rank, n5, world))
                sweep_frames(rank, mine, world)
                torch.cuda.synchronize()
                dist.barrier()
                dt5 = time.perf_counter() - t0
                redo = torch.tensor([0 if ctx.synchronize() == capi.C5_OK else 1], dtype=torch.int64, device=rdev)
                dist.all_reduce(redo)
                if int(redo.item()) == 0:
                    break
            t5 = torch.tensor([dt5], dtype=torch.float64, device=rdev)
            dist.all_reduce(t5, op=dist.ReduceOp.MAX)
            dt5 = float(t5.item())
            # one GPU: rank 0 alone renders as many frames of the same sweep as one rank just did
            host_barrier()
            one5 = None
            if rank == 0:
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    sweep_frames(0, per_rank, 1)
                    torch.cuda.synchronize()
                    one5 = (time.perf_counter() - t0) * 1e3 / per_rank
                    if ctx.synchronize() == capi.C5_OK:
                        break
            host_barrier()
            sweep.update(kind=args.sweep, solids=bool(args.solids), stride=1, k=0)
            ctx.set_option("view_cache", 1 if args.sweep == "D" else 0)
            put_solids(False)
            del whole
        rays5 = base_res[0] * base_res[1]
        config5 = {"what": f"BASELINE config 5: {n5}-frame -D sweep (D = k / 180) at {base_res[0]}x{base_res[1]} with the Roche lobe "
                           f"(130 560 tets) and the accretor sphere (522 242 tets) resident, whole frames dealt to {world} GPUs "
                           "(frame k on rank k mod N), grid + solids uploaded once, no exchange; images stay in HBM",
                   "value": round(rays5 * n5 / dt5 / 1e6, 2), "unit": "Mrays/s", "frames": n5, "frames_per_s": round(n5 / dt5, 1),
                   "ms_per_frame_job": round(dt5 * 1e3 / n5, 4), "incomplete_frames_reported": int(redo.item()),
                   "view_cache": "on (the product's default): the view of a donor sweep is fixed, so transformed vertices, cell records "
                                 "and entry lists are built twice and reused by every later frame; only the lobe is turned and rastered per frame"}
        if rank == 0 and one5:
            config5.update(one_gpu_ms_per_frame=round(one5, 4), one_gpu_frames_per_s=round(1e3 / one5, 1),
                           speedup_vs_one_gpu=round(one5 / (dt5 * 1e3 / n5), 3))
        with torch.cuda.stream(stream):  # back to the headline's layout (the native host below uses its own contexts)
            lay_out_rows(res_x, res_y, False)

    # The C++ host on the same workload: rank 0 runs it as a child process on all N GPUs while the other ranks
    # idle at the barrier below (their GPUs are free: nothing of this job is running on them).
    native = None
    if not args.no_native and not args.solids and args.sweep == "none" and args.workload == "c3":
        torch.cuda.synchronize()
        host_barrier()
        if rank == 0:
            # The protocol of these legs does not follow --steps (BENCH_r03: 20 timed frames behind 5 warm ones gave 86
            # frames/s where 120-frame runs give 130-150): >= 100 timed frames per round behind >= 20 warm ones (through
            # the same path: writer thread, files), several rounds; the line carries the whole run, the fastest and the
            # median round (ms_per_frame / _min / _median), and README / DESIGN quote exactly what this prints.
            k = 100
            if world == 1:
                # sweep_files: the reference's real workload end to end (utility/rotate_traces.py:16-21 renders 1 500
                # frames of a -Y sweep to files): every timed frame rendered, copied to pinned memory, deflated and
                # written as a zlib .vti by the writer thread while the next frames render
                variants = [("one_gpu", []),
                            ("sweep_files", ["--bench_files", "--sweep_step", "0.00555556", "--bench", "100", "--bench_rounds", "3",
                                             "--bench_warmup", "20", "-j", str(usable_cpus())])]
            else:
                variants = [("rows_host", ["--split", "rows", "--exchange", "host"]),
                            ("rows_host_tiles", ["--split", "rows", "--exchange", "host", "--row_layout", "tiles"]),
                            ("rows_rccl", ["--split", "rows", "--exchange", "rccl"]),
                            ("rows_rccl_tiles", ["--split", "rows", "--exchange", "rccl", "--row_layout", "tiles"]),
                            ("rows_p2p", ["--split", "rows", "--exchange", "p2p"]),
                            ("frames", ["--split", "frames"])]
            native = native_course_bench(xyz, cells, alpha, q, res_x, res_y, world, k, 40, variants)
            if world > 1 and not args.no_configs:
                c4x, c4y = (int(v) for v in args.config4_res.lower().split("x"))
                native["config4"] = native_course_bench(xyz, cells, alpha, q, c4x, c4y, world, max(10, k // 2), 10,
                                                        [("rows_host", ["--split", "rows", "--exchange", "host"]),
                                                         ("rows_rccl", ["--split", "rows", "--exchange", "rccl"])])
        host_barrier()

    if rank == 0:
        frames = args.steps * (world if frame_parallel else 1)
        rays = res_x * res_y
        ms_per_step = elapsed * 1e3 / args.steps
        value = rays * frames / elapsed / 1e6
        # roofline of walk_composite on rank 0 (its own rows)
        S_rank, P_rank = stats["segments"], n_local * res_x
        alg_bytes = S_rank * B_SEG_SURVEY + P_rank * B_PIX
        secs = walk_ms * 1e-3
        pmc = load_roofline_counters() if (world == 1 and args.workload == "c3" and base_res == (2400, 1800) and
                                            not args.solids and args.sweep == "none") else None
        hbm_bytes = pmc.get("hbm_bytes_per_launch") if pmc else None
        achieved = hbm_bytes / secs / 1e9 if (hbm_bytes and secs > 0) else None
        # The counters and the phase clock were collected by separate profiling runs (scripts/collect_pmc.sh,
        # scripts/stamp_walk.py) and committed; they describe the kernels of that moment.  The hash of the kernel
        # sources they were taken from travels with them: if the sources have changed since, say so.
        from course5_amd.build import kernel_source_hash
        src_hash = kernel_source_hash()
        pmc_stale = bool(pmc) and pmc.get("source_hash") != src_hash
        phases = (pmc or {}).get("phases")
        lim = (pmc or {}).get("limiter") or {}
        bound_names = {"chain": "latency chain of a wavefront-step at 8 wavefronts per SIMD (no unit saturated: vector pipes ~55 % weighted by "
                                "instruction cost, the CU's scalar unit ~65 %, LDS ~60 %, HBM 13-16 %; profiles/r04_walk_isa.md)",
                       "valu": "vector-instruction issue (VALU quad-cycles, no MFMA)", "salu": "scalar-instruction issue", "lds": "LDS",
                       "l2": "L2 requests", "hbm": "hbm"}
        roofline = {
            # `bound` says what binds the kernel by the committed instruments: the loop's instructions by class at their
            # measured issue cost against the counters (profiles/<round>_walk_isa.md, _pmc.md; `limiter` holds the units'
            # busy shares over the launch and while the wavefront slots are full), `phases` where a wavefront-step spends
            # its cycles, `timeline` how the launch fills and drains.  HBM is NOT it (13 %: the 128 MB of per-view records
            # are served from L2 / Infinity Cache): achieved / peak / frac stay the HBM-side figures the contract asks
            # for — HBM bytes per launch over the live kernel time against 8 TB/s.
            "bound": bound_names.get(lim.get("name"), "hbm"),
            "nearest_roofline": "hbm", "kernel": "walk_composite",
            "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
            "traffic": hbm_bytes,
            "traffic_source": (pmc or {}).get("source"),
            "kernel_ms": round(walk_ms, 4), "launches": walk_launches,
            "kernel_ms_sampling": "HIP events around every 4th launch of the timed region when it has 200 steps or more, else around every launch (two events cost 6 us of a frame)",
            "kernel_ms_rocprofv3": (pmc or {}).get("kernel_ms_rocprofv3"),
            "limiter": (pmc or {}).get("limiter"),
            "units": (pmc or {}).get("units"),
            "phases": phases, "timeline": (pmc or {}).get("timeline"),
            "pmc_round": (pmc or {}).get("round"), "pmc_source_hash": (pmc or {}).get("source_hash"),
            "source_hash": src_hash, "pmc_stale": pmc_stale,
            "contract": {
                "what": "SURVEY.md section 8(d): algorithmic bytes (S x 144 B + P x 8 B) / kernel time. Not a fraction "
                        "of a hardware limit: the per-view records (128 MB) are re-read from L2 / Infinity Cache, "
                        "HBM sees `traffic` bytes per launch",
                "algorithmic_bytes_per_launch": alg_bytes,
                "achieved_gbs": round(alg_bytes / secs / 1e9, 1) if secs > 0 else None,
                "x_hbm_peak": round(alg_bytes / secs / 1e9 / HBM_PEAK_GBS, 3) if secs > 0 else None,
                "bytes_per_segment": B_SEG_SURVEY, "bytes_per_pixel": B_PIX,
                "segments_per_launch": S_rank, "pixels_per_launch": P_rank,
                "record_bytes_per_segment": B_SEG_RECORD,
            },
        }
        if world == 1:
            metric = "Mrays/sec at 2400x1800 on 1M-tet grid"
        elif frame_parallel:
            metric = (f"Mrays/sec, sweep of whole {res_x}x{res_y} frames dealt to {world} GPUs (frame k -> GPU k mod N, "
                      f"no exchange): STRONG scaling of the sweep")
        elif args.scaling == "weak":
            metric = (f"Mrays/sec, WEAK scaling: {base_res[0]}x{base_res[1]} rays per GPU, one {res_x}x{res_y} image "
                      f"split by rows over {world} GPUs (an N-fold value here is not north_star's fixed-frame speed-up)")
        else:
            metric = (f"Mrays/sec at {res_x}x{res_y} on 1M-tet grid, STRONG scaling: the N = 1 frame split by rows over {world} GPUs "
                      f"(north_star's 1 -> 8 GPU clause)")
        out = {
            "metric": metric, "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "none" if world == 1 else ("strong" if frame_parallel else args.scaling),
            "vs_baseline": None, "dtype": "f64", "data": "synthetic", "retries": retries_total,
            **({"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "ranks": seen} if world > 1 else {}),
            "steadying": {"batches_of_50_frames": len(steady), "walk_ms_per_batch": [round(v, 4) for v in steady]},
            "config": {"workload": f"{args.workload}: Kuhn box 55^3 = {cells.shape[0]} tets, {xyz.shape[0]} points, "
                                   f"jitter 0.1h, alpha~U[0,4) Q~U[0,1) seed 1234; {res_x}x{res_y}; view -X 0.1 -Y 0.07; "
                                   f"alpha_limit 2.5; {'Roche lobe + sphere' if args.solids else 'no solids'}" if args.workload == "c3" else
                                   f"{args.workload}: {cells.shape[0]} tets; {res_x}x{res_y}",
                       "parallelism": "single GPU" if world == 1 else
                                      (f"whole frames dealt round-robin to {world} ranks, grid replicated, no exchange" if frame_parallel else
                                       (f"row tiles of {TILE_ROWS} rows dealt cyclically to {world} ranks" if blocks is None else
                                        f"{world} contiguous row blocks balanced by measured segments per row, then by the GPU times the ranks took for them, at multiples of 8 rows "
                                        f"(rows per rank {[n for _, n in blocks]})") +
                                       ", grid replicated, one RCCL exchange per frame to rank 0"),
                       "segments_per_frame": S_total if not frame_parallel else stats["segments"],
                       "pixels_per_frame": P_total if not frame_parallel else res_x * res_y,
                       "sweep": args.sweep, "solids": bool(args.solids),
                       "view_cache": ("on: the view of a -D sweep is fixed (per-view data built twice, then reused)" if args.sweep == "D"
                                      else "off: every timed frame does its whole per-view setup"),
                       "walk_kernel": ("walk_composite_lds<3, 0, true, 14, SMALLEXP = true>: the instantiation without the general exp, taken while "
                                       "min(alpha limit, largest alpha) x longest cell edge < 1/8 (this workload: 0.10; with --alpha_limit 3.0 on the "
                                       "same grid 0.12).  The general instantiation (SMALLEXP = false: wave-uniform choice between the short series "
                                       "and the full exp) runs the same frame about 3 % slower (0.474 against 0.460 ms, round 3)"),
                       "depth_split": "0 (per frame): this frame's jobs fill the wavefront slots 2.5 times over, its rays stay whole",
                       "rays_per_gpu": P_total // world,
                       "frames_timed": frames},
            "roofline": roofline,
        }
        if sharded and head_one_gpu_ms:
            # the same image on ONE GPU (rank 0 alone, same run): what value(N) is a speed-up over
            out["one_gpu"] = {"ms_per_frame": round(head_one_gpu_ms, 4), "value": round(rays / head_one_gpu_ms / 1e3, 2),
                              "frames": k_ref, "what": f"rank 0 rendering the whole {res_x}x{res_y} frame alone, the other ranks idle"}
            out["speedup_vs_one_gpu"] = round(head_one_gpu_ms / ms_per_step, 3)
            pred = predicted_scaling(res_x, res_y, world)
            if pred:
                out["predicted"] = pred
        if config4 is not None:
            pred = predicted_scaling(*(int(v) for v in args.config4_res.lower().split("x")), world)
            if pred:
                config4["predicted"] = pred
            out["config4"] = config4
        if config5 is not None:
            out["config5"] = config5
        if host_image is not None:
            out["value_host_image"] = host_image
        if native is not None:
            out["native_host"] = {"what": "the C++ host `course --bench` (one process, one c5_context per GPU) on the same grid, "
                                          "view and image; frames delivered to pinned host memory (sweep_files: written as zlib "
                                          ".vti files, end to end); Mrays/s in mrays_per_s",
                                  **native}
        if world == 1 and not args.no_cpu_baseline:
            sres = tuple(int(v) for v in args.cpu_sample_res.lower().split("x")) if args.cpu_sample_res else (res_x, res_y)
            out["cpu_baseline"], _ = cpu_baseline(xyz, cells, alpha, q, rots, sres)
            if not args.no_cpu_reference:
                ref_line = cpu_baseline_reference(xyz, cells, alpha, q, rots, sres)
                if ref_line:
                    out["cpu_baseline_reference"] = ref_line
        print(json.dumps(out), flush=True)

    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
