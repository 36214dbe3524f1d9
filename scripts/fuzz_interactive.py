"""Randomised parity run for the INTERACTIVE path (render_interactive, render.c:465-574): one persistent carver per library,
a random sequence of resizes inside and beyond the cached map (both directions, shrinking and enlarging), flattens, with masks,
rigidity and the energy function drawn per case; after EVERY call the getters, the image and the dumped map must agree.

    python scripts/fuzz_interactive.py [seconds] [seed]
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
o = L.oracle_api()
e = L.engine_api()
e.lib.lqrhip_set_update_mode.argtypes = [ctypes.c_int]
fails = FC.Failures(e.lib)
n = ok_steps = err_breaks = 0
while budget.more(n):
    big = rng.random() < 0.3
    w, h = (int(rng.integers(900, 1500)), int(rng.integers(60, 160))) if big else (int(rng.integers(24, 260)), int(rng.integers(16, 160)))
    ch = int(rng.integers(1, 5))
    img = [D.noise, D.photo_like, D.flat_blocks][int(rng.integers(0, 3))](w, h, int(rng.integers(0, 1 << 30)), channels=ch)
    kw = dict(nrg_func=int(rng.integers(0, 7)), switch_freq=int(rng.choice([0, 1, 2, 3])), res_order=int(rng.integers(0, 2)),
              enl_step=float(rng.choice([150.0, 120.0, 200.0])))
    if rng.random() < 0.2:
        kw.update(rigidity=float(rng.choice([1.0, 8.0])))
    if rng.random() < 0.15:
        kw.update(delta_x=int(rng.choice([2, 3])))
    mk = dict(pres=D.ellipse_mask(w, h), disc=D.band_mask(w, h, w // 5, w // 3)) if rng.random() < 0.2 else {}
    mode = int(rng.choice([-1, 0, 2]))
    e.lib.lqrhip_set_update_mode(mode)
    steps = []
    cw, chh = w, h
    for _ in range(int(rng.integers(2, 7))):
        if rng.random() < 0.2:
            steps.append(("f",))
            continue
        span = 40 if not big else 25
        nw = int(np.clip(cw + rng.integers(-span, span // 2 + 1), 4, int(cw * 1.4)))
        nh = int(np.clip(chh + (rng.integers(-20, 11) if rng.random() < 0.4 else 0), 4, int(chh * 1.4)))
        steps.append(("r", nw, nh)); cw, chh = nw, nh
    what = "%dx%d ch%d %s%s mode %d steps %s" % (w, h, ch, kw, " +masks" if mk else "", mode, steps)
    try:
        cs = [H.init_carver(api, img, steps[0][1] if steps[0][0] == "r" else w, steps[0][2] if steps[0][0] == "r" else h, **kw, **mk)[0] for api in (o, e)]
        for st in steps:
            rets = [c.resize(st[1], st[2]) if st[0] == "r" else c.flatten() for c in cs]
            assert rets[0] == rets[1], "return values %s at %s" % (rets, st)
            if rets[0] != L.LQR_OK:
                err_breaks += 1
                break
            ok_steps += 1
            assert cs[0].getters() == cs[1].getters(), "getters at %s" % (st,)
            assert np.array_equal(cs[0].read_image(), cs[1].read_image()), "image at %s" % (st,)
            va, vb = cs[0].vmap_dump(), cs[1].vmap_dump()
            assert va["depth"] == vb["depth"] and np.array_equal(va["data"], vb["data"]), "map at %s" % (st,)
        for c in cs:
            c.destroy()
    except Exception as ex:
        fails.record(n, what, ex)
    n += 1
e.lib.lqrhip_set_update_mode(-1)
FC.summary("interactive fuzz", n, budget, fails, seed, " (%d calls compared, %d sequences ended by an error both libraries returned)" % (ok_steps, err_breaks))
sys.exit(1 if fails.total else 0)
