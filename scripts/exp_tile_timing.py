"""per-phase cycle accounting of k_dp_tile_p<UPDATE> (library built with make -C gimp-lqr-plugin_amd EXTRA=-DLQR_TIMING), one image: the middle tile's two waves"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
ge._import_package()
from gimp_lqr_plugin_amd import binding as L
import bench as B
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
eng = L.engine_api(); lib = eng.lib
lib.lqrhip_init()
dev = torch.device("cuda", 0)
images = B.make_images(1, W, H, 100, dev); torch.cuda.synchronize()
c = L.Carver(eng, images[0].cpu().numpy()).configure(switch_freq=2, enl_step=1.5)
assert c.resize(W - 40, H) == L.LQR_OK
out = (C.c_ulonglong * 32)()
lib.lqrhip_tile_timing.argtypes = [C.POINTER(C.c_ulonglong)]
assert lib.lqrhip_tile_timing(out) == 0
a = np.array(out[:], dtype=np.int64).reshape(2, 16)
names = ["loop/idle", "poll", "landed", "rows", "handover", "barrier", "holdback", "stores", "issue", "-", "total"]
nb = (H + 15) // 16
print("%dx%d: %d batches of 16 rows per tile, last launch, middle tile (cycles; per-batch in brackets for the wave's own %d batches)" % (W, H, nb, nb // 2))
for q in range(2):
    print("wave %d  " % q + "  ".join("%s %d (%d)" % (names[i], a[q, i], a[q, i] // max(nb // 2, 1)) for i in range(11) if names[i] != "-"))
