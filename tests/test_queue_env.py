"""GPU_MAX_HW_QUEUES and the library (csrc/lqr_shim.hip lqrhip_on_load, lqrhip_sub_batches): the 4-stream schedule of large
lock-step groups needs 8 hardware queues, the HIP runtime reads the variable once when it comes up, and a host like the
plug-in does not know it exists.  Loaded before the runtime is up the library sets it; a host's own value is kept; loaded
into a process whose runtime is already up (it can no longer be changed) the library stays on one stream."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gimp-lqr-plugin_amd", "liblqr-hip.so")
PROBE = """
import ctypes, sys
libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p
%s
lib = ctypes.CDLL(%r)
v = libc.getenv(b"GPU_MAX_HW_QUEUES")
print("RESULT", v.decode() if v else "unset", lib.lqrhip_sub_batches(64), lib.lqrhip_sub_batches(16))
"""


def probe(env_value, before=""):
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    if env_value is not None:
        env["GPU_MAX_HW_QUEUES"] = env_value
    r = subprocess.run([sys.executable, "-c", PROBE % (before, LIB)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][0].split()[1:]


def test_library_sets_the_queue_count_when_the_host_did_not():
    assert probe(None) == ["8", "4", "1"]            # 64 images: 4 streams; 16 images: one


@pytest.mark.parametrize("value,streams", [("4", "1"), ("16", "4")])
def test_a_hosts_own_value_is_kept(value, streams):
    assert probe(value) == [value, streams, "1"]


@pytest.mark.gpu
def test_runtime_already_up_means_one_stream():
    """the host initialised HIP (with the default 4 queues) before it loaded the library: setting the variable now would only
    make lqrhip_sub_batches believe in queues that are not there (4 streams on shared queues are 30 % slower than one)"""
    assert probe(None, before="import torch; torch.cuda.init(); torch.zeros(1, device='cuda')") == ["unset", "1", "1"]


def test_bench_asks_for_more_queues_than_its_own_streams_need():
    """bench.py's N > 1 runs have RCCL in the process, whose communicator brings streams of its own: with 8 hardware queues the four
    sub-batch streams shared queues with them (400 k instead of 575 k at world size 1 under torch.distributed.run, round 6,
    profiles/r06/q_queues_under_torchrun.txt).  bench.py must export at least 12 before torch is imported; a caller's own value is kept."""
    code = "import os, sys; sys.path.insert(0, %r); import bench; print('RESULT', os.environ.get('GPU_MAX_HW_QUEUES'), 'torch' in sys.modules)" % ROOT
    for given, want in ((None, None), ("9", "9")):
        env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "LQR_BENCH_NO_QUEUE_ENV")}
        if given:
            env["GPU_MAX_HW_QUEUES"] = given
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        value, torch_loaded = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][0].split()[1:]
        assert torch_loaded == "False"          # the variable is in place before the first HIP call of the process
        assert (value == want) if want else int(value) >= 12, value
