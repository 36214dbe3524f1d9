"""Randomised parity run (engine vs oracle through the C ABI) for a fixed wall-clock budget, cycling
through the three update_mmap paths (cases: tests/fuzz_cases.py).

    python scripts/fuzz_parity.py [seconds] [seed] [verbose] [general]

`general` (4th argument) bends every case towards the tiled kernels' general instantiations: delta_x 2 in half the
cases, a rigidity mask with rigidity in half, and only the engine's own choice of update kernel.
"""
import ctypes
import sys
import time

sys.path.insert(0, "tests")
import numpy as np

import fuzz_cases as F
import fuzz_common as FC
import harness as H
import lqr_ctypes as L

budget = FC.Budget(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0)      # FUZZ_COUNT=n: exactly n cases, no wall-clock exit
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
verbose = len(sys.argv) > 3 and sys.argv[3] not in ("0", "")
general = len(sys.argv) > 4
# FUZZ_ONLY=i,j,...: draw every case (so the random stream stays the same) but run only these indices, each
# FUZZ_REPEAT times, in every update mode listed in FUZZ_MODES (default: the one the index would have had)
import os
only = set(int(x) for x in os.environ.get("FUZZ_ONLY", "").split(",") if x)
repeat = int(os.environ.get("FUZZ_REPEAT", "1"))
force_modes = [int(x) for x in os.environ.get("FUZZ_MODES", "").split(",") if x]
# FUZZ_EXTRAS=1: the plug-in's other switches too, drawn from a second stream (the cases themselves stay the same): seam-map
# output, attached mask layers resized along, LqR-back, discard masks kept on enlargement, the enlargement step
extras = np.random.default_rng(seed + 1000003) if os.environ.get("FUZZ_EXTRAS") else None
import datasets as D
rng = np.random.default_rng(seed)
o = L.oracle_api()
e = L.engine_api()
e.lib.lqrhip_set_update_mode.argtypes = [ctypes.c_int]
if os.environ.get("LQR_VP"):             # the parallel backtrack forced for every case (default: images of 1400 rows and more only)
    e.lib.lqrhip_set_vpath_mode.argtypes = [ctypes.c_int, ctypes.c_int]; e.lib.lqrhip_set_vpath_mode(int(os.environ["LQR_VP"]), 0)
fails = FC.Failures(e.lib)
n = 0
while budget.more(n):
    img, nw, nh, kw, what = F.draw_case(rng)
    name, mode = list(F.MODES.items())[n % 3]
    if n % 4 == 3:                       # round 5: every fourth case on k_band_levels (the cases themselves stay what the seed made them)
        name, mode = "levels", 5
    if general:
        name, mode = "auto", -1
        h_, w_ = img.shape[:2]
        if rng.random() < 0.5:
            kw["delta_x"] = 2
        else:
            kw.pop("delta_x", None)
        if rng.random() < 0.5:
            kw["rigmask"] = D.top_half_mask(w_, h_) if rng.random() < 0.5 else D.ellipse_mask(w_, h_)
            kw["rigidity"] = float(rng.choice([0.5, 3.0, 40.0]))
        elif rng.random() < 0.5:
            kw["rigidity"] = float(rng.choice([1.0, 8.0]))
        what += " general:%s" % {k: v for k, v in kw.items() if k in ("delta_x", "rigidity")} + (" +rigmask" if "rigmask" in kw else "")
    if extras is not None:
        ex = dict(output_seams=bool(extras.random() < 0.3), resize_aux_layers=bool(extras.random() < 0.3), scaleback=bool(extras.random() < 0.25),
                  no_disc_on_enlarge=bool(extras.random() < 0.7), enl_step=float(extras.choice([150.0, 110.0, 200.0])))
        kw.update(ex)
        what += " extras:%s" % {k: v for k, v in ex.items() if v not in (False, 150.0)}
    if only and n not in only:
        n += 1
        if n > max(only):
            break
        continue
    what = name + " " + what
    if verbose:
        print("case", n, what, flush=True)
    a = None
    for m in (force_modes or [mode]):
        e.lib.lqrhip_set_update_mode(m)
        for rep in range(repeat):
            try:
                if a is None:
                    a = H.run_case(o, img, nw, nh, **kw)
                b = H.run_case(e, img, nw, nh, **kw)
                H.assert_same(a, b, what)
                if only:
                    print("ok   case %d mode %d rep %d" % (n, m, rep), flush=True)
            except Exception as ex:
                fails.record("%d mode %d rep %d" % (n, m, rep), what, ex)
    n += 1
e.lib.lqrhip_set_update_mode(-1)
FC.summary("fuzz", n, budget, fails, seed)
sys.exit(1 if fails.total else 0)
