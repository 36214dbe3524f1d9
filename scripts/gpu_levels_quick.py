"""k_band_levels (update mode 5) against the oracle on a few shapes and slot counts, then its launch time next to the other
forms of update_mmap on one 4K image.   python scripts/gpu_levels_quick.py"""
import ctypes as C, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, "tests")
import numpy as np
import lqr_ctypes as L, datasets as D, harness as H
o = L.oracle_api(); e = L.engine_api(); lib = e.lib
for f in ("lqrhip_set_update_mode", "lqrhip_set_band_levels", "lqrhip_set_sub_batches"): getattr(lib, f).argtypes = [C.c_int]
lib.lqrhip_band_levels_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
def stats():
    st = (C.c_ulonglong * 8)(); lib.lqrhip_band_levels_stats(st, 1); return [int(x) for x in st[:4]]
if os.environ.get("LQR_LV_DBG"):
    lib.lqrhip_band_levels_debug.argtypes = [C.c_int]; lib.lqrhip_band_levels_debug(int(os.environ["LQR_LV_DBG"]))
bad = 0
cases = [("photo 300x160", D.photo_like(300, 160, 73), 260, 160, {}),
         ("noise 1200x200", D.noise(1200, 200, 5), 1150, 200, dict(switch_freq=0)),
         ("photo 1400x700 both", D.photo_like(1400, 700, 11), 1340, 680, {}),
         ("flat 900x300 rigidity", D.flat_blocks(900, 300, 4), 850, 300, dict(rigidity=4.0)),
         ("noise 2500x120 every-seam switch", D.noise(2500, 120, 9), 2460, 120, dict(switch_freq=1000)),
         ("null energy + masks 276x80", D.flat_blocks(276, 80, 4), 216, 80, dict(nrg_func=L.LQR_EF_NULL, pres=D.ellipse_mask(276, 80))),
         ("tiny 40x9", D.noise(40, 9, 2), 30, 9, {})]
for name, img, nw, nh, kw in cases:
    ref = H.run_case(o, img, nw, nh, **kw)
    for P in (1, 2, 3, 6, 12, 16):
        lib.lqrhip_set_update_mode(5); lib.lqrhip_set_band_levels(P)
        try:
            got = H.run_case(e, img, nw, nh, **kw)
            H.assert_same(ref, got, name)
            print("ok   %-34s P=%2d  stats[coll,sync,proc,idle]=%s" % (name, P, stats()), flush=True)
        except Exception as ex:
            bad += 1
            print("FAIL %-34s P=%2d  %s  last_error=%r stats=%s" % (name, P, str(ex)[:200], lib.lqrhip_last_error(), stats()), flush=True)
lib.lqrhip_set_update_mode(-1); lib.lqrhip_set_band_levels(-1)
print("levels quick: %d failures" % bad)
sys.exit(1 if bad else 0)
