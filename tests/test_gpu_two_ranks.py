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
