"""How long does the persistent tiled keep-rule sweep (k_dp_tile_p<UPDATE>) take when the chip holds ~770 of its tiles -- the
size a band-restricted variant for 64 x 4K would have (64 images x 12 tiles of 64 columns, or x 6 tiles of 128)?
    python scripts/exp_tiles.py [images] [width] [px]"""
import ctypes as C, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, "tests")
import numpy as np
import lqr_ctypes as L
n, W, px = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 1
bt = int(sys.argv[5]) if len(sys.argv) > 5 else -1
rsv = int(sys.argv[6]) if len(sys.argv) > 6 else -1
H, seams = 2160, 24
eng = L.engine_api(); lib = eng.lib
for f in ("lqrhip_set_update_mode", "lqrhip_set_sub_batches", "lqrhip_set_dp_persistent_px", "lqrhip_set_band_tiles", "lqrhip_set_band_tiles_reserve"): getattr(lib, f).argtypes = [C.c_int]
lib.lqrhip_set_band_tiles(bt); lib.lqrhip_set_band_tiles_reserve(rsv)
lib.lqrhip_set_update_mode(mode); lib.lqrhip_set_sub_batches(1); lib.lqrhip_set_dp_persistent_px(px)
rng = np.random.default_rng(1)
cs = [L.Carver(eng, rng.integers(0, 256, (H, W, 4), dtype=np.uint8)).configure(switch_freq=0) for _ in range(n)]
assert L.resize_batch(eng, cs, W - 4, H) == 1        # warm-up (allocates the second planes)
lib.lqrhip_prof_reset(); lib.lqrhip_prof_enable(1)
assert L.resize_batch(eng, cs, W - 4 - seams, H) == 1
lib.lqrhip_prof_enable(0)
out = []
for k in ("vpath", "carve", "emap_update", "band_update", "dp_update_tiled", "dp_update", "dp_sweep"):
    ms, cnt, by = C.c_double(0), C.c_longlong(0), C.c_double(0)
    lib.lqrhip_prof_get(k.encode(), C.byref(ms), C.byref(cnt), C.byref(by))
    if cnt.value: out.append("%s %.1f us x %d" % (k, ms.value * 1e3 / cnt.value, cnt.value))
st = (C.c_ulonglong * 8)(); lib.lqrhip_band_tiles_stats(st, 1)
print("mode %d bt %d rsv %d stats [uncov, abort, woken, norsv] %s:" % (mode, bt, rsv, list(st)[:8]), end=" ")
print("%d images of %dx%d, %d px per lane (%d tiles): %s" % (n, W, H, px, n * ((W + (64 * px - 32 * px) - 1) // (64 * px - 32 * px)), "; ".join(out)), flush=True)
for c in cs: c.destroy()
