"""Randomised parity run for the multi-CU band update k_band_levels (update mode 5): engine vs oracle through the C ABI with
1..16 slots per image forced -- one slot stops at the first level with three active tiles (hand-over to the full-width sweep),
many slots leave most of them idle.  Also compares the DP planes after the last incremental update bit for bit.
(Round 4's k_band_tiles, which this script was written for, is gone; the case mix and the seeds' draws are unchanged.)
    python scripts/fuzz_levels.py [seconds] [seed]"""
import ctypes, os, sys, time
sys.path.insert(0, "tests")
import numpy as np
import datasets as D, fuzz_cases as F, fuzz_common as FC, harness as H, lqr_ctypes as L

budget = FC.Budget(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)       # FUZZ_COUNT=n: exactly n cases, no wall-clock exit
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = int(os.environ.get("FUZZ_ONLY", "-1"))            # replay: draw every case, run only this one (FUZZ_REPEAT times)
repeat = int(os.environ.get("FUZZ_REPEAT", "1"))
rng = np.random.default_rng(seed)
o, e = L.oracle_api(), L.engine_api()
lib = e.lib
lib.lqrhip_set_update_mode.argtypes = [ctypes.c_int]
lib.lqrhip_set_band_levels.argtypes = [ctypes.c_int]
fails = FC.Failures(lib)
if os.environ.get("LQR_LV_DBG"):        # k_band_levels' experiment switches (4 no near copy, 8 an image's slots on different XCDs)
    lib.lqrhip_band_levels_debug.argtypes = [ctypes.c_int]; lib.lqrhip_band_levels_debug(int(os.environ["LQR_LV_DBG"]))
n = 0
try:
    while budget.more(n):
        img, nw, nh, kw, what = F.draw_case(rng)
        kw.pop("delta_x", None); kw.pop("rigmask", None)          # the plain kernels' domain
        if rng.random() < 0.5:
            kw["switch_freq"] = 0
        T = int(rng.choice([1, 2, 3, 4, 6, 8, 12]))
        rsv = int(rng.integers(-1, T))             # reserve tiles among them (-1: the engine's third)
        if only >= 0 and n != only:
            n += 1
            if n > only:
                break
            continue
        T = int([1, 2, 3, 4, 6, 8, 12, 16][(T * 7 + rsv + 1) % 8])          # slots per image (derived from what the seed drew: same cases)
        lib.lqrhip_set_update_mode(5); lib.lqrhip_set_band_levels(T)
        planes = kw.get("switch_freq") == 0 and nh == img.shape[0] and nw < img.shape[1]
        for rep in range(repeat if only >= 0 else 1):
            try:
                if planes:
                    o.lqrx_set_debug(1); e.lqrx_set_debug(1)
                ca, _ = H.init_carver(o, img, nw, nh, **kw); cb, _ = H.init_carver(e, img, nw, nh, **kw)
                ra, rb = ca.resize(nw, nh), cb.resize(nw, nh)
                assert ra == rb == 1, "resize returned %s (oracle) / %s (engine)" % (ra, rb)
                va, vb = ca.vmap_dump(), cb.vmap_dump()
                assert np.array_equal(va["data"], vb["data"]), "seam maps"
                assert np.array_equal(ca.read_image(), cb.read_image()), "pixels"
                if planes:
                    (ea, ma, da), (eb, mb, db) = ca.debug_snapshot(), cb.debug_snapshot()
                    assert np.array_equal(ea, eb) and np.array_equal(ma, mb) and np.array_equal(da[1:], db[1:]), "DP planes"
                ca.destroy(); cb.destroy()
            except Exception as ex:
                fails.record(n, "slots=%d %s" % (T, what), ex)
            finally:
                o.lqrx_set_debug(0); e.lqrx_set_debug(0)
        n += 1
finally:
    lib.lqrhip_set_update_mode(-1); lib.lqrhip_set_band_levels(-1)
FC.summary("levels fuzz", n, budget, fails, seed)
sys.exit(1 if fails.total else 0)
