"""Where a k_band_update_tw launch goes, alone and under the 4-stream schedule's load (a timing build: make -C gimp-lqr-plugin_amd -j8
EXTRA=-DLQR_TIMING BUILD=build_timing OUT=liblqr-hip-timing.so): cycles per 16-row batch and wave of image 0's workgroup, LAST launch.
    python scripts/exp_tw_timing.py [images] [sub-batches]"""
import ctypes as C, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, "tests")
import numpy as np
import lqr_ctypes as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sub = int(sys.argv[2]) if len(sys.argv) > 2 else 1
W, H = 3840, 2160
eng = L.Api(os.path.join(os.path.dirname(L.ENGINE_LIB), "liblqr-hip-timing.so"), ""); lib = eng.lib
for f in ("lqrhip_set_update_mode", "lqrhip_set_sub_batches"): getattr(lib, f).argtypes = [C.c_int]
lib.lqrhip_set_update_mode(0); lib.lqrhip_set_sub_batches(sub)
import datasets as D
base = [D.photo_like(W, H, 100 + i) for i in range(min(n, 4))]
imgs = [base[i] if i < 4 else np.roll(base[i % 4], 37 * i, axis=1) for i in range(n)]
cs = [L.Carver(eng, im).configure(switch_freq=0) for im in imgs]
lib.lqrhip_prof_reset(); lib.lqrhip_prof_enable(1)
ret = L.resize_batch(eng, cs, W - 24, H) if n > 1 else cs[0].resize(W - 24, H)
assert ret == L.LQR_OK, lib.lqrhip_last_error()
lib.lqrhip_prof_enable(0)
ms, nl, by = C.c_double(0), C.c_longlong(0), C.c_double(0)
lib.lqrhip_prof_get(b"band_update", C.byref(ms), C.byref(nl), C.byref(by))
print("k_band_update_tw by HIP events: %.1f us per launch over %d launches" % (ms.value * 1e3 / max(nl.value, 1), nl.value))
out = (C.c_ulonglong * 64)()
assert lib.lqrhip_band_tw_timing(out) == 0
a = np.array(out[:], dtype=np.float64).reshape(8, 8)
nb = (H + 15) // 16
names = ["landed", "batch", "barrier", "to issue", "issue", "kernel"]
print("%d images on %d stream(s): cycles per 16-row batch (%d batches), wave (slot = wave %% 4; the waves of a slot alternate batches)" % (n, sub, nb))
print("%-6s" % "wave" + "".join("%10s" % x for x in names))
for w in range(8):
    print("%-6d" % w + "".join("%10.0f" % (a[w, i] / nb) for i in range(5)) + "%10.0f" % (a[w, 5] / nb))
