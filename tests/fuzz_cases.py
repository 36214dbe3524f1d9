"""Randomised parity cases (engine vs oracle through the C ABI), deterministic per seed.  Shared by
tests/test_fuzz_gpu.py (fixed seed list, -m gpu) and scripts/fuzz_parity.py (wall-clock budget).
Sizes are drawn so that the band kernel's window (896 columns) is smaller than the image often
enough to exercise re-centring and the hand-over; delta_x, rigidity, masks, energy function,
side-switch frequency, resize order, channels and dataset all vary."""
import numpy as np

import datasets as D

MODES = {"auto": -1, "band": 0, "band-mw": 2, "levels": 5}


def draw_case(rng, small=False):
    kind = int(rng.integers(0, 4))
    if small:
        kind = 2 if kind < 2 else 3
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
    if rng.random() < 0.25:        # wider seam steps, mostly without rigidity: seams wander (k_emap_update's sample window)
        kw.update(delta_x=int(rng.choice([0, 2, 3, 5, 10, 16])))
    if rng.random() < 0.15:
        kw.update(pres=D.ellipse_mask(w, h), disc=D.band_mask(w, h, w // 5, w // 3))
    if rng.random() < 0.05:
        kw.update(rigmask=D.top_half_mask(w, h), rigidity=kw.get("rigidity", 2.0))
    what = "%s %dx%d ch%d -> %dx%d %s" % (gen.__name__, w, h, ch, w + dw, h + dh,
                                          {k: v for k, v in kw.items() if k not in ("pres", "disc", "rigmask")})
    if "pres" in kw:
        what += " +masks"
    if "rigmask" in kw:
        what += " +rigmask"
    return img, w + dw, h + dh, kw, what
