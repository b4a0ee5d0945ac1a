#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the render hot path at 2400x1800 on the ~1M-tet synthetic grid.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one frame through the whole hot path on data already resident in HBM: view transform ->
per-cell records -> boundary entry raster -> walk_composite -> fp32 image in device memory
(N > 1: each rank renders its cyclic row tiles, then one RCCL gather of the strips to rank 0 and
the reassembly there).  Rank 0 prints ONE JSON line.

value       whole-job Mrays/s = res_x * res_y * K / (max over ranks of the timed region) / 1e6
roofline    dominant kernel walk_composite: algorithmic bytes per launch (SURVEY.md §8(d):
            S * 144 B + P * 8 B) / its average duration, measured live with HIP events on the
            stream the kernel runs on (c5_walk_kernel_ms)
cpu_baseline  the CPU oracle (own restatement of the reference algorithm, OpenMP) timed on this
            box's host cores on a bounded sample of the same workload (rank 0, N = 1 only).
            It is the checker being timed as a baseline, never the product path.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from course5_amd import capi, meshgen as mg, sharding  # noqa: E402
from course5_amd.pipeline import FramePipeline, gather_row_costs  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
HBM_ACHIEVABLE_GBS = 6290.0  # measured float4 copy, same table
B_SEG_SURVEY = 144           # 16 cell->vertex + 16 adjacency + 96 vertices + 16 scalars
B_SEG_RECORD = 160           # what walk_composite actually loads per step: 128 B CellRecord + 32 B CellOptics
B_PIX = 8                    # 2 x fp32 store per pixel
TILE_ROWS = 16


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--workload", default="c3", help="c3 (998 250 tets, default), c2, kuhnN")
    p.add_argument("--res", default="2400x1800", help="image size of the N = 1 workload")
    p.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                   help="N > 1: weak = per-GPU rays stay at --res (image grows by sqrt(N) per side; N = 4 is "
                        "BASELINE config 4, 4800x3600); strong = the same --res frame split over N GPUs")
    p.add_argument("--tile", type=int, default=-1, help="wavefront tile shape override (0: 64x1, 1: 16x4, 2: 8x8)")
    p.add_argument("--lds-stage", type=int, default=-1, help="override: 1 = LDS-staged walk kernel, 0 = direct loads")
    p.add_argument("--pipeline", type=int, default=-1, help="override: overlap the next frame's setup with the walk (1) or not (0)")
    p.add_argument("--overlap-setup", type=int, default=-1, help="override: entry lists beside build_records (1) or serial (0)")
    p.add_argument("--own-stream", action="store_true", help="run on the context's own (high priority) stream")
    p.add_argument("--backend", default="nccl", help="nccl (= RCCL, default); gloo only to rehearse N > 1 on one GPU")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample-res", default="1200x900")
    p.add_argument("--pipeline-depth", type=int, default=2, help="N > 1: frames whose gather may be in flight")
    p.add_argument("--sharding", choices=["blocks", "cyclic"], default="blocks",
                   help="N > 1: contiguous cost-balanced row blocks (default) or cyclic 16-row tiles")
    p.add_argument("--row-base-cost", type=float, default=0.5,
                   help="blocks: fixed cost per pixel in segment units added when balancing")
    return p.parse_args()


def cpu_baseline(xyz, cells, alpha, q, rots, sample_res):
    """Oracle ("port") on the host cores, bounded sample: same grid and view, reduced image."""
    from oracle.pyoracle import Oracle
    cores = min(len(os.sched_getaffinity(0)), 32)  # 32 = MAX_NUMBER_OF_THREADS (config.hpp:39)
    rx, ry = sample_res
    r = Oracle("port").render(xyz, cells, alpha, q, rots, rx, ry, mg.REFERENCE_BOUNDS, threads=cores)
    ctor, binning, resolve = (float(v) for v in r["timing_ms"])
    span = ctor + binning + resolve
    return {
        "value": rx * ry / span / 1e3, "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": f"same grid and view at {rx}x{ry} ({r['segments']} segments); span = pixel grid + binning + "
                  f"resolve (the reference's own timed span, main.cpp:126-130) = {span:.0f} ms; "
                  f"binning + resolve only = {binning + resolve:.0f} ms",
        "value_bin_resolve_only": rx * ry / (binning + resolve) / 1e3,
    }, r["segments"]


def load_traffic():
    """HBM bytes per walk_composite launch from the committed rocprofv3 PMC summary, if any."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if os.environ.get("C5_BENCH_ONE_DEVICE"):  # rehearsal only: every rank on GPU 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    res_x, res_y = (int(v) for v in args.res.lower().split("x"))
    base_res = (res_x, res_y)
    if world > 1 and args.scaling == "weak":
        res_x, res_y = int(round(res_x * world ** 0.5)), int(round(res_y * world ** 0.5))
    xyz, cells, alpha, q = mg.workload(args.workload)
    rots = mg.view_rotations(**mg.BENCH_VIEW)

    stream = torch.cuda.Stream(device=dev)
    ctx = capi.Context(local_rank)
    if args.pipeline >= 0:
        ctx.set_option("pipeline", args.pipeline)
    ctx.upload_grid(xyz, cells, alpha, q)
    ctx.set_image(res_x, res_y, mg.REFERENCE_BOUNDS)
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

    def render(strip):
        ctx.render_device(strip.data_ptr())

    blocks = None
    with torch.cuda.stream(stream):
        if world > 1 and args.sharding == "cyclic":
            ctx.set_row_tiles(TILE_ROWS, rank, world)
        elif world > 1:
            # one probe frame on equal blocks measures segments per row; every rank then derives the
            # same cost-balanced contiguous blocks (pixels are independent: plane.cpp:161-169)
            eq = sharding.equal_blocks(res_y, world)
            ctx.set_row_range(*eq[rank])
            ctx.set_option("row_costs", 1)
            probe = torch.zeros((eq[rank][1], res_x, 2), dtype=torch.float32, device=dev)
            render(probe)
            while ctx.synchronize() == capi.C5_RETRY:
                render(probe)
            costs = gather_row_costs(ctx.row_costs(), eq, rank, world, dev if args.backend == "nccl" else torch.device("cpu"))
            blocks = sharding.balanced_blocks(costs, world, base_cost=res_x * args.row_base_cost)
            ctx.set_option("row_costs", 0)
            ctx.set_row_range(*blocks[rank])
            del probe
    n_local = ctx.local_rows
    pipe = FramePipeline(res_x, res_y, rank, world, dev, depth=args.pipeline_depth, tile_rows=TILE_ROWS, blocks=blocks,
                         host_staging=args.backend != "nccl")

    with torch.cuda.stream(stream):
        # warm-up (also lets the entry buffer reach its size: C5_RETRY means "render again")
        for k in range(max(args.warmup, 1)):
            pipe.step(render)
            if k < 2:
                pipe.drain()
                while ctx.synchronize() == capi.C5_RETRY:
                    pipe.step(render)
                    pipe.drain()
        pipe.drain()
        ctx.synchronize()
        stats = ctx.stats()
        ctx.walk_kernel_ms(reset=True)

        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            pipe.step(render)
        pipe.drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if ctx.synchronize() != capi.C5_OK:
            raise SystemExit("frame had to be re-rendered inside the timed region; run with more warmup")
        walk_ms, walk_launches = ctx.walk_kernel_ms(reset=True)

    rdev = dev if args.backend == "nccl" else torch.device("cpu")
    el = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
    seg = torch.tensor([stats["segments"], n_local * res_x], dtype=torch.int64, device=rdev)
    wk = torch.tensor([walk_ms], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(seg, op=dist.ReduceOp.SUM)
        dist.all_reduce(wk, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    S_total, P_total = int(seg[0].item()), int(seg[1].item())

    if rank == 0:
        rays = res_x * res_y
        ms_per_step = elapsed * 1e3 / args.steps
        value = rays * args.steps / elapsed / 1e6
        # roofline of walk_composite on rank 0 (its own rows): algorithmic bytes per launch / duration
        S_rank, P_rank = stats["segments"], n_local * res_x
        alg_bytes = S_rank * B_SEG_SURVEY + P_rank * B_PIX
        achieved = alg_bytes / (walk_ms * 1e-3) / 1e9 if walk_ms > 0 else 0.0
        traffic = load_traffic()
        roofline = {
            "bound": "hbm", "kernel": "walk_composite", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": (traffic or {}).get("hbm_bytes_per_launch"),
            "traffic_source": (traffic or {}).get("source"),
            "kernel_ms": round(walk_ms, 4), "launches": walk_launches,
            "algorithmic_bytes_per_launch": alg_bytes,
            "bytes_per_segment": B_SEG_SURVEY, "bytes_per_pixel": B_PIX,
            "segments_per_launch": S_rank, "pixels_per_launch": P_rank,
            "record_bytes_per_segment": B_SEG_RECORD,
            "achieved_record_accounting": round((S_rank * B_SEG_RECORD + P_rank * B_PIX) / (walk_ms * 1e-3) / 1e9, 1)
            if walk_ms > 0 else 0.0,
            "frac_of_achievable_6290": round(achieved / HBM_ACHIEVABLE_GBS, 4),
            "note": "algorithmic bytes exceed HBM traffic: the per-view records (160 MB) are served by L2 / "
                    "Infinity Cache, so frac > 1 is possible; see traffic for the measured HBM bytes",
        }
        out = {
            "metric": "Mrays/sec at 2400x1800 on 1M-tet grid", "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: Kuhn box 55^3 = {cells.shape[0]} tets, {xyz.shape[0]} points, "
                                   f"jitter 0.1h, alpha~U[0,4) Q~U[0,1) seed 1234; {res_x}x{res_y}; view -X 0.1 -Y 0.07; "
                                   f"alpha_limit 2.5; no solids" if args.workload == "c3" else
                                   f"{args.workload}: {cells.shape[0]} tets; {res_x}x{res_y}",
                       "parallelism": "single GPU" if world == 1 else
                                      (f"row tiles of {TILE_ROWS} rows dealt cyclically to {world} ranks" if blocks is None else
                                       f"{world} contiguous row blocks balanced by measured segments per row "
                                       f"(rows per rank {[n for _, n in blocks]})") +
                                      ", grid replicated, one RCCL gather per frame to rank 0",
                       "segments_per_frame": S_total, "pixels_per_frame": P_total,
                       "rays_per_gpu": P_total // world,
                       "scaling_note": None if world == 1 else
                       (f"weak: {base_res[0]}x{base_res[1]} rays per GPU, image {res_x}x{res_y}" if args.scaling == "weak"
                        else f"strong: one {res_x}x{res_y} frame split over {world} GPUs")},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            sres = tuple(int(v) for v in args.cpu_sample_res.lower().split("x"))
            out["cpu_baseline"], _ = cpu_baseline(xyz, cells, alpha, q, rots, sres)
        print(json.dumps(out), flush=True)

    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
