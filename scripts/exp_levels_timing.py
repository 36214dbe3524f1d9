"""Where a k_band_levels launch goes (a timing build: make -C gimp-lqr-plugin_amd -j8 EXTRA=-DLQR_TIMING BUILD=build_timing
OUT=liblqr-hip-timing.so): per phase, per slot of image 0 and wave, of the LAST launch.
    python scripts/exp_levels_timing.py [images] [slots]"""
import ctypes as C, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, "tests")
import numpy as np
import lqr_ctypes as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
P = int(sys.argv[2]) if len(sys.argv) > 2 else 12
W, H = 3840, 2160
eng = L.Api(os.path.join(os.path.dirname(L.ENGINE_LIB), os.environ.get("LQR_TIMING_LIB", "liblqr-hip-timing.so")), ""); lib = eng.lib
for f in ("lqrhip_set_update_mode", "lqrhip_set_sub_batches", "lqrhip_set_band_levels"): getattr(lib, f).argtypes = [C.c_int]
lib.lqrhip_set_update_mode(5); lib.lqrhip_set_sub_batches(int(os.environ.get("LQR_EXP_STREAMS", "1"))); lib.lqrhip_set_band_levels(P)
rng = np.random.default_rng(5)
imgs = [rng.integers(0, 256, (H, W, 4), dtype=np.uint8) for _ in range(n)]
for im in imgs: im[..., 3] = 255
cs = [L.Carver(eng, im).configure(switch_freq=0) for im in imgs]
ret = L.resize_batch(eng, cs, W - 12, H) if n > 1 else cs[0].resize(W - 12, H)
assert ret == L.LQR_OK, lib.lqrhip_last_error()
out = (C.c_ulonglong * 320)()
assert lib.lqrhip_band_levels_timing(out) == 0
a = np.array(out[:], dtype=np.float64).reshape(16, 2, 10)
names = ["poll", "set+choice", "wait partner", "stores", "loads", "row above", "32 rows", "publish", "LDS barrier", "kernel"]
nlev = (H + 31) // 32
print("%d images, %d slots: ticks of s_memtime (100 MHz: 10 ns) per LEVEL (a wave takes every other one), slot.wave; %d levels; kernel = whole launch in us" % (n, P, nlev))
print("%-10s" % "slot.wave" + "".join("%13s" % x for x in names))
for s in range(min(P, 16)):
    for q in range(2):
        print("%-10s" % ("%d.%d" % (s, q)) + "".join("%13.1f" % (a[s, q, i] / nlev) for i in range(9)) + "%13.1f" % (a[s, q, 9] / 100.0))
