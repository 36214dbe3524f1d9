"""Randomised parity run (engine vs oracle through the C ABI) for a fixed wall-clock budget, cycling
through the three update_mmap paths.  Sizes are drawn so that the band kernel's window (896 columns)
is smaller than the image often enough to exercise re-centring and the hand-over.

    python scripts/fuzz_parity.py [seconds] [seed]
"""
import ctypes
import sys
import time

sys.path.insert(0, "tests")
import numpy as np

import datasets as D
import harness as H
import lqr_ctypes as L

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
verbose = len(sys.argv) > 3
rng = np.random.default_rng(seed)
o = L.oracle_api()
e = L.engine_api()
e.lib.lqrhip_set_update_mode.argtypes = [ctypes.c_int]
modes = {"auto": -1, "band": 0, "band-mw": 2}
t_end = time.time() + budget
n = fails = 0
while time.time() < t_end:
    kind = rng.integers(0, 4)
    if kind == 0:      # wide and low: window < image, fast oracle
        w, h = int(rng.integers(900, 2600)), int(rng.integers(40, 200))
    elif kind == 1:    # tall: wide bands, re-centring, hand-over
        w, h = int(rng.integers(950, 1500)), int(rng.integers(500, 1100))
    elif kind == 2:    # small
        w, h = int(rng.integers(8, 300)), int(rng.integers(4, 200))
    else:
        w, h = int(rng.integers(300, 1000)), int(rng.integers(100, 500))
    gen = [D.noise, D.photo_like, D.flat_blocks][int(rng.integers(0, 3))]
    ch = int(rng.integers(1, 5))
    img = gen(w, h, int(rng.integers(0, 1 << 30)), channels=ch)
    dw = int(rng.integers(-min(w - 2, 90), 60))
    dh = int(rng.integers(-min(h - 2, 40), 20)) if rng.random() < 0.4 else 0
    kw = dict(nrg_func=int(rng.integers(0, 7)), switch_freq=int(rng.choice([0, 1, 2, 3, 9])), res_order=int(rng.integers(0, 2)))
    if rng.random() < 0.15:
        kw.update(rigidity=float(rng.choice([1.0, 8.0, 100.0])))
    if rng.random() < 0.15:
        kw.update(pres=D.ellipse_mask(w, h), disc=D.band_mask(w, h, w // 5, w // 3))
    name, mode = list(modes.items())[n % 3]
    e.lib.lqrhip_set_update_mode(mode)
    what = "%s %dx%d ch%d -> %dx%d %s %s" % (gen.__name__, w, h, ch, w + dw, h + dh, name, kw if "pres" not in kw else "masks")
    if verbose:
        print("case", n, what, flush=True)
    try:
        a = H.run_case(o, img, w + dw, h + dh, **kw)
        b = H.run_case(e, img, w + dw, h + dh, **kw)
        H.assert_same(a, b, what)
    except AssertionError as ex:
        fails += 1
        print("FAIL", what, str(ex)[:200], flush=True)
    n += 1
e.lib.lqrhip_set_update_mode(-1)
print("fuzz: %d cases, %d failures, seed %d" % (n, fails, seed), flush=True)
sys.exit(1 if fails else 0)
