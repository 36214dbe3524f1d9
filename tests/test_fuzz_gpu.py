"""Randomised parity (-m gpu): 240 seeded cases of tests/fuzz_cases.py, engine vs oracle through the C
ABI, cycling through the three update_mmap paths.  Every case compares seam maps, pixels, getters,
aux layers, dumped maps and progress events bit for bit (harness.assert_same)."""
import ctypes

import numpy as np
import pytest

import fuzz_cases as F
import harness as H

pytestmark = pytest.mark.gpu

SEEDS = [11, 12, 13, 14, 15, 16, 17, 18]
CASES_PER_SEED = 30


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_seed(oracle, engine, seed):
    engine.lib.lqrhip_set_update_mode.argtypes = [ctypes.c_int]
    rng = np.random.default_rng(seed)
    try:
        for n in range(CASES_PER_SEED):
            # every third case large (band window < image); the others small, so the whole file stays ~2 min
            img, nw, nh, kw, what = F.draw_case(rng, small=(n % 3 != 0))
            name, mode = list(F.MODES.items())[(n + n // 3) % len(F.MODES)]      # the cases visit all the update modes
            engine.lib.lqrhip_set_update_mode(mode)
            a = H.run_case(oracle, img, nw, nh, **kw)
            b = H.run_case(engine, img, nw, nh, **kw)
            H.assert_same(a, b, "seed %d case %d %s %s" % (seed, n, name, what))
    finally:
        engine.lib.lqrhip_set_update_mode(-1)
