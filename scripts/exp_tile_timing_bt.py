"""Where a k_band_tiles launch goes (a timing build: make -C gimp-lqr-plugin_amd EXTRA=-DLQR_TIMING):
cycles per phase, per tile slot of image 0 and wave, of the LAST launch.   python scripts/exp_tile_timing_bt.py [images]"""
import ctypes as C, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, "tests")
import numpy as np
import lqr_ctypes as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W, H = 3840, 2160
eng = L.engine_api(); lib = eng.lib
for f in ("lqrhip_set_update_mode", "lqrhip_set_sub_batches", "lqrhip_set_band_tiles"): getattr(lib, f).argtypes = [C.c_int]
lib.lqrhip_set_update_mode(4); lib.lqrhip_set_sub_batches(1); lib.lqrhip_set_band_tiles(12)
rng = np.random.default_rng(1)
cs = [L.Carver(eng, rng.integers(0, 256, (H, W, 4), dtype=np.uint8)).configure(switch_freq=0) for _ in range(n)]
assert (L.resize_batch(eng, cs, W - 12, H) if n > 1 else cs[0].resize(W - 12, H)) == 1
t = (C.c_ulonglong * 320)()
assert lib.lqrhip_band_tiles_timing(t) == 0
a = np.array(t[:]).reshape(16, 2, 10)
print("slot wave: receive compute rest barrier wait_partner store issue | active blocks | total  (cycles, 100 MHz wall clock x21 = core cycles?)")
for s in range(12):
    for q in range(2):
        r = a[s, q]
        print("%2d %d: %8d %8d %8d %8d %8d %8d %8d | %3d | %8d" % (s, q, *r[:7], r[7], r[8]))
