"""Randomised parity run for lock-step BATCHES (lqrx_carver_resize_batch): groups of 2-9 images of one size with different
contents and common parameters, engine as one batch vs oracle image by image; sub-batch streams, update modes and the
opt-in seam-round forms vary with the case.

    python scripts/fuzz_batch.py [seconds] [seed]          (FUZZ_COUNT=n: exactly n cases instead of the wall-clock budget)
"""
import ctypes
import os
import sys
import time

sys.path.insert(0, "tests")
import numpy as np

import datasets as D
import fuzz_common as FC
import harness as H
import lqr_ctypes as L

budget = FC.Budget(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0)      # FUZZ_COUNT=n: exactly n cases, no wall-clock exit
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
rng_lv = np.random.default_rng(seed + 500009)        # round 5: a third of the cases on k_band_levels, drawn from a stream of its own (the cases stay the same)
o = L.oracle_api()
e = L.engine_api()
lib = e.lib
for f in ("lqrhip_set_update_mode", "lqrhip_set_sub_batches"):
    getattr(lib, f).argtypes = [ctypes.c_int]
fails = FC.Failures(lib)
if os.environ.get("LQR_LV_DBG"):        # k_band_levels' experiment switches (4 no near copy, 8 an image's slots on different XCDs)
    lib.lqrhip_band_levels_debug.argtypes = [ctypes.c_int]; lib.lqrhip_band_levels_debug(int(os.environ["LQR_LV_DBG"]))
n = 0
while budget.more(n):
    kind = int(rng.integers(0, 3))
    if kind == 0:
        w, h = int(rng.integers(900, 2400)), int(rng.integers(40, 160))
    elif kind == 1:
        w, h = int(rng.integers(300, 1000)), int(rng.integers(100, 400))
    else:
        w, h = int(rng.integers(16, 300)), int(rng.integers(8, 150))
    nimg = int(rng.integers(2, 10))
    ch = int(rng.integers(1, 5))
    gens = [D.noise, D.photo_like, D.flat_blocks]
    imgs = [gens[int(rng.integers(0, 3))](w, h, int(rng.integers(0, 1 << 30)), channels=ch) for _ in range(nimg)]
    dw = int(rng.integers(-min(w - 2, 60), 40))
    dh = int(rng.integers(-min(h - 2, 30), 15)) if rng.random() < 0.4 else 0
    kw = dict(nrg_func=int(rng.integers(0, 7)), switch_freq=int(rng.choice([0, 1, 2, 3, 9])), res_order=int(rng.integers(0, 2)))
    if rng.random() < 0.2:
        kw.update(rigidity=float(rng.choice([1.0, 8.0])))
    if rng.random() < 0.2:
        kw.update(delta_x=int(rng.choice([2, 3])))
    masks = rng.random() < 0.2
    mode = int(rng.choice([-1, 0, 1, 2, 4, 4]))       # (4 was k_band_tiles, removed in round 6: those draws run k_band_levels now)
    mode = 5 if mode == 4 else mode
    sub = int(rng.choice([1, 1, 2, 3]))
    if rng_lv.random() < 0.34:
        mode = 5
    what = "%d x %dx%d ch%d -> %dx%d %s%s mode %d sub %d" % (nimg, w, h, ch, w + dw, h + dh, kw, " +masks" if masks else "", mode, sub)
    lib.lqrhip_set_update_mode(mode); lib.lqrhip_set_sub_batches(sub)
    mk = dict(pres=D.ellipse_mask(w, h), disc=D.band_mask(w, h, w // 5, w // 3)) if masks else {}
    try:
        cs = [H.init_carver(e, im, w + dw, h + dh, **kw, **mk)[0] for im in imgs]
        assert L.resize_batch(e, cs, w + dw, h + dh) == L.LQR_OK, "resize_batch failed: %s" % lib.lqrhip_last_error().decode()
        for i, (im, c) in enumerate(zip(imgs, cs)):
            ref = H.run_case(o, im, w + dw, h + dh, **kw, **mk)
            got_img, got_map = c.read_image(), c.vmap_dump()["data"]
            assert np.array_equal(got_map, ref["vmap"]["data"]), "image %d: seam maps differ" % i
            assert np.array_equal(got_img, ref["image"]), "image %d: pixels differ" % i
        for c in cs:
            c.destroy()
    except Exception as ex:
        # is it a state of the process or a passing event?  the same batch once more
        again = "not rerun"
        try:
            cs2 = [H.init_carver(e, im, w + dw, h + dh, **kw, **mk)[0] for im in imgs]
            ok2 = L.resize_batch(e, cs2, w + dw, h + dh) == L.LQR_OK
            for im, c in zip(imgs, cs2):
                ref = H.run_case(o, im, w + dw, h + dh, **kw, **mk)
                ok2 = ok2 and np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"]) and np.array_equal(c.read_image(), ref["image"])
                c.destroy()
            again = "rerun of the same batch: " + ("ok" if ok2 else "differs again")
        except Exception as ex2:
            again = "rerun raised %s" % str(ex2)[:100]
        fails.record(n, what + " [" + again + "]", ex)
    n += 1
lib.lqrhip_set_update_mode(-1); lib.lqrhip_set_sub_batches(0)
FC.summary("batch fuzz", n, budget, fails, seed)
sys.exit(1 if fails.total else 0)
