"""Where a fused seam round's time goes (a -DLQR_FUSED_TIMING build): when the workers published each 32-row chunk of
image 0 and when the band update's wave 0 asked for / got it.   python scripts/exp_fused_timing.py [images] [sub_batches]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
import __graft_entry__ as ge
ge._import_package()
from gimp_lqr_plugin_amd import binding as L
import bench as B
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
W, H = 3840, 2160
eng = L.engine_api(); lib = eng.lib
lib.lqrhip_init()
lib.lqrhip_set_sub_batches.argtypes = [C.c_int]; lib.lqrhip_set_sub_batches(sb)
dev = torch.device("cuda", 0)
images = B.make_images(n, W, H, 100, dev); torch.cuda.synchronize()
cs = [L.Carver(eng, images[i].cpu().numpy()).configure(switch_freq=2, enl_step=1.5) for i in range(n)]
assert L.resize_batch(eng, cs, W - 60, H) == L.LQR_OK
buf = (C.c_ulonglong * 512)()
lib.lqrhip_fused_timing.argtypes = [C.POINTER(C.c_ulonglong)]
assert lib.lqrhip_fused_timing(buf) == 0
z = np.array(buf[:], dtype=np.int64).reshape(4, 128)
t0 = z[3, 0]
print("%d images, %d sub-batches: band update start 0, end %.1f us" % (n, sb, (z[3, 1] - t0) / 100.0))
print("chunk  published  asked  got   (us after the band update started)")
for c in range(0, 68, 3):
    print("%4d %9.1f %8s %8s" % (c, (z[0, c] - t0) / 100.0, "%.1f" % ((z[1, c] - t0) / 100.0) if z[1, c] else "-", "%.1f" % ((z[2, c] - t0) / 100.0) if z[2, c] else "-"))
