"""Replay of tests/test_build_variants.py's first case (photo 300x160 -> 260x140, the engine's own choice of kernels) N times against the
oracle, on whatever build LQR_HIP_LIB names: looks for the rare mismatch soak 3 of round 5 saw once on the default-scheduler build.
    python scripts/repro_buildvariant.py [N] [w h nw nh]"""
import ctypes, os, sys
sys.path.insert(0, "tests")
import numpy as np
import datasets as D, harness as H, lqr_ctypes as L
import time
secs, n = None, 100                                           # runs, or -- with a trailing "s" -- seconds
if len(sys.argv) > 1 and sys.argv[1].endswith("s"):
    secs, n = float(sys.argv[1][:-1]), 1 << 30
elif len(sys.argv) > 1:
    n = int(sys.argv[1])
t_end = time.time() + (secs or 1e18)
w, h, nw, nh = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (300, 160, 260, 140)
o = L.oracle_api(); e = L.engine_api()
img = D.photo_like(w, h, 73)
ref = H.run_case(o, img, nw, nh)
bad = 0
runs = 0
for i in range(n):
    if time.time() > t_end:
        break
    runs += 1
    got = H.run_case(e, img, nw, nh)
    try:
        H.assert_same(ref, got, "run %d" % i)
    except AssertionError as ex:
        bad += 1
        d = np.argwhere(ref["vmap"]["data"] != got["vmap"]["data"])
        lv = sorted(set(int(ref["vmap"]["data"][tuple(p)]) for p in d) | set(int(got["vmap"]["data"][tuple(p)]) for p in d))
        print("MISMATCH run %d: %s; levels involved %s; last_error=%r" % (i, str(ex)[:160], lv[:12], e.lib.lqrhip_last_error()), flush=True)
        again = []
        for _ in range(3):
            g2 = H.run_case(e, img, nw, nh)
            again.append("ok" if np.array_equal(ref["vmap"]["data"], g2["vmap"]["data"]) and np.array_equal(ref["image"], g2["image"]) else "differs")
        print("   the same case three more times: %s" % again, flush=True)
print("repro: %d runs, %d mismatches, lib %s" % (runs, bad, os.environ.get("LQR_HIP_LIB", "default")))
