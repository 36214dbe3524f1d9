"""The progress-publishing carve (k_seam_work) and the band update that waits for it (k_band_update_tw_f), -m gpu.

DESIGN.md section 4.14: opt-in forms of the seam round (lqrhip_set_fused 1: one stream, 2: two streams, 3: k_carve_pub).  They are slower
than the default and stay in the library as the measured record of that experiment -- and must stay bit-identical to the
oracle: seeded cases of tests/fuzz_cases.py in child processes (LQRHIP_FUSED is read when the library is loaded), and a
batch on two streams compared with the default path in this process."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

import datasets as D
import lqr_ctypes as L

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["1", "2", "3"])
def test_seeded_cases(mode):
    env = dict(os.environ, LQRHIP_FUSED=mode, FUZZ_COUNT="150", GPU_MAX_HW_QUEUES="16")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "600", "5150"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " 0 failures" in r.stdout, (r.stdout[-3000:], r.stderr[-1000:])


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_batch_matches_default_path(engine, mode):
    lib = engine.lib
    lib.lqrhip_set_fused.argtypes = [ctypes.c_int]
    lib.lqrhip_set_update_mode.argtypes = [ctypes.c_int]
    w, h, n = 1500, 300, 6
    imgs = [D.photo_like(w, h, 900 + i, channels=4) for i in range(n)]
    outs = {}
    try:
        lib.lqrhip_set_update_mode(0)                     # the band kernel whatever the batch size
        for m in (0, mode):
            lib.lqrhip_set_fused(m)
            cs = [L.Carver(engine, im).configure() for im in imgs]
            assert L.resize_batch(engine, cs, w - 70, h) == L.LQR_OK
            outs[m] = [(c.read_image(), c.vmap_dump()["data"]) for c in cs]
            for c in cs:
                c.destroy()
    finally:
        lib.lqrhip_set_fused(0)
        lib.lqrhip_set_update_mode(-1)
    for a, b in zip(outs[0], outs[mode]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
