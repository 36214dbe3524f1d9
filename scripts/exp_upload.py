"""where the upload phase goes: lqr_carver_new (device allocation + H2D) vs lqr_carver_init (working planes), first time
(cold block cache) and second time (blocks recycled); read-out rate"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, "tests")
import numpy as np
import lqr_ctypes as L
eng = L.engine_api()
lib = eng.lib
n, W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 3840, 2160
imgs = [np.random.default_rng(i).integers(0, 256, (H, W, 4), dtype=np.uint8) for i in range(4)]
for rnd in range(6):
    bufs = [L._malloc_copy(imgs[i % 4]) for i in range(n)]
    lib.lqrhip_device_sync()
    t0 = time.perf_counter()
    ps = [eng.lqr_carver_new(bufs[i], W, H, 4) for i in range(n)]
    lib.lqrhip_device_sync()
    t1 = time.perf_counter()
    for p in ps:
        assert eng.lqr_carver_init(p, 1, 0.0) == 1
    lib.lqrhip_device_sync()
    t2 = time.perf_counter()
    gb = n * W * H * 4 / 1e9
    print("round %d: new %.1f ms (%.1f GB/s), init %.1f ms" % (rnd, (t1 - t0) * 1e3, gb / (t1 - t0), (t2 - t1) * 1e3), flush=True)
    outs = [np.empty((H, W, 4), np.uint8) for _ in range(n)]
    for tag in ("fresh", "touched"):
        t3 = time.perf_counter()
        for p, o in zip(ps, outs):
            assert eng.lqrx_carver_read_image(p, o.ctypes.data) == 1
        t4 = time.perf_counter()
        print("   read-out into %s buffers %.1f ms (%.1f GB/s)" % (tag, (t4 - t3) * 1e3, gb / (t4 - t3)), flush=True)
    for p in ps:
        eng.lqr_carver_destroy(p)
