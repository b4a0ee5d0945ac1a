"""N = 2 rehearsal of the multi-GPU frame pipeline on the ONE GPU of the test box: two ranks (gloo, both
on device 0, strips staged through the host) render their row tiles / blocks through FramePipeline —
side-stream reassembly, two gathers in flight — and every reassembled frame must equal the
single-context render bit for bit (tests/two_rank_check.py).  The RCCL transport itself cannot be
exercised on one GPU; everything around it is."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("layout,extra", [("cyclic", {}), ("blocks", {}),
                                          # the configuration of the round-1 wrong frame (two frame slots per context,
                                          # blocks), started from a pool far too small: frames are re-rendered, never wrong
                                          ("blocks", {"C5_PIPELINE": "1", "C5_ENTRY_POOL": "100"})],
                         ids=["cyclic", "blocks", "blocks-two-slots-starved-pool"])
def test_two_ranks_reassemble_every_frame(layout, extra):
    env = dict(os.environ, C5_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    import socket
    with socket.socket() as sock:  # a port nobody holds right now
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", port,
                        os.path.join(ROOT, "tests", "two_rank_check.py"), layout],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PASS" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    if extra:
        import re
        again = [int(m) for m in re.findall(r"rank \d+: (\d+) frame\(s\) rendered again", r.stdout)]
        assert len(again) == 2 and min(again) > 0, r.stdout[-2000:]  # the starved pool did cost re-renders


def test_bench_two_ranks_reports_the_baseline_configs():
    """`python bench.py --gpus 2` started by ONE plain command (bench.py spawns its ranks itself, as
    a child `torch.distributed.run` job, before it has touched the GPU; here: gloo, both ranks on the one GPU, a small
    grid and small images): the headline is the N = 1 frame split by rows (strong scaling) beside its own one-GPU time, followed by
    BASELINE config 4 (a larger frame split by rows) and config 5 (a -D sweep with the solids, whole frames dealt to
    the ranks), each with the one-GPU time of the same work measured by rank 0 in the same run."""
    import json
    env = dict(os.environ, C5_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "6", "--warmup", "2", "--backend", "gloo", "--workload", "kuhn12",
                        "--res", "400x300", "--config4-res", "800x600", "--config5-frames", "8", "--no-native", "--no-steady"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["scaling"] == "strong" and b["steps"] == 6 and b["value"] > 0
    assert b["rccl_ranks"] == 2 and sorted(r_["rank"] for r_ in b["ranks"]) == [0, 1] and len({r_["pid"] for r_ in b["ranks"]}) == 2
    assert "400x300" in b["metric"] and "STRONG" in b["metric"]
    assert b["one_gpu"]["ms_per_frame"] > 0 and b["speedup_vs_one_gpu"] > 0
    assert abs(b["speedup_vs_one_gpu"] - b["one_gpu"]["ms_per_frame"] / b["ms_per_step"]) < 0.01
    c4 = b["config4"]
    assert c4["value"] > 0 and c4["one_gpu_ms_per_frame"] > 0 and c4["speedup_vs_one_gpu"] > 0
    assert sum(c4["rows_per_rank"]) == 600 and len(c4["rows_per_rank"]) == 2
    c5 = b["config5"]
    assert c5["frames"] == 8 and c5["frames_per_s"] > 0 and c5["speedup_vs_one_gpu"] > 0 and c5["incomplete_frames_reported"] == 0
    assert b["roofline"]["source_hash"]
