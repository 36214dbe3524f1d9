"""Randomised parity run (engine vs oracle through the C ABI) for a fixed wall-clock budget, cycling
through the three update_mmap paths (cases: tests/fuzz_cases.py).

    python scripts/fuzz_parity.py [seconds] [seed]
"""
import ctypes
import sys
import time

sys.path.insert(0, "tests")
import numpy as np

import fuzz_cases as F
import harness as H
import lqr_ctypes as L

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
verbose = len(sys.argv) > 3
rng = np.random.default_rng(seed)
o = L.oracle_api()
e = L.engine_api()
e.lib.lqrhip_set_update_mode.argtypes = [ctypes.c_int]
t_end = time.time() + budget
n = fails = 0
while time.time() < t_end:
    img, nw, nh, kw, what = F.draw_case(rng)
    name, mode = list(F.MODES.items())[n % 3]
    e.lib.lqrhip_set_update_mode(mode)
    what = name + " " + what
    if verbose:
        print("case", n, what, flush=True)
    try:
        a = H.run_case(o, img, nw, nh, **kw)
        b = H.run_case(e, img, nw, nh, **kw)
        H.assert_same(a, b, what)
    except AssertionError as ex:
        fails += 1
        print("FAIL", what, str(ex)[:200], flush=True)
    n += 1
e.lib.lqrhip_set_update_mode(-1)
print("fuzz: %d cases, %d failures, seed %d" % (n, fails, seed), flush=True)
sys.exit(1 if fails else 0)
