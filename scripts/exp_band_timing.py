"""Dump the per-wave cycle accounting of k_band_update_td (library built with -DLQR_BAND_TIMING) after a short 64 x 4K run.
   python scripts/exp_band_timing.py BAND_KERNEL [nimg] [seams]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
ge._import_package()
from gimp_lqr_plugin_amd import binding as L
import bench as B
bk = int(sys.argv[1]); nimg = int(sys.argv[2]) if len(sys.argv) > 2 else 64; seams = int(sys.argv[3]) if len(sys.argv) > 3 else 40
eng = L.engine_api(); lib = eng.lib
lib.lqrhip_init()
lib.lqrhip_set_band_kernel.argtypes = [C.c_int]; lib.lqrhip_set_band_kernel(bk)
lib.lqrhip_set_band_variant.argtypes = [C.c_int]; lib.lqrhip_set_band_variant(int(os.environ.get('BAND_VARIANT', '0'))); rep = int(os.environ.get('BAND_REPEAT', '1')); lib.lqrhip_set_band_repeat.argtypes = [C.c_int]; lib.lqrhip_set_band_repeat(rep)
lib.lqrhip_set_update_mode.argtypes = [C.c_int]; lib.lqrhip_set_update_mode(0)
dev = torch.device("cuda", 0)
W, H = 3840, 2160
images = B.make_images(nimg, W, H, 100, dev); torch.cuda.synchronize()
ptrs = [images[i].data_ptr() for i in range(nimg)]
carvers = [L.Carver(eng, np.zeros((H, W, 4), np.uint8)).configure(switch_freq=2, enl_step=1.5) for _ in range(nimg)]
for it in range(2):
    assert L.reload_device_batch(eng, carvers, ptrs) == L.LQR_OK
    r = L.resize_batch(eng, carvers, W - seams, H) if nimg > 1 else carvers[0].resize(W - seams, H)
    if r != L.LQR_OK: print('RESIZE FAILED', r); break
if bk == 3:
    hist = (C.c_ulonglong * 64)()
    lib.lqrhip_ls_hist.argtypes = [C.POINTER(C.c_ulonglong)]
    assert lib.lqrhip_ls_hist(hist) == 0
    hh = list(hist)
    print("ls kernel: launches", hh[32], "overflowed", hh[33], "batches", hh[34], "active slot-batches", hh[35], "rows handed over", hh[36])
    print("predicted range width histogram (x32 px):", hh[:32])
    print('fail record: n, j, slot, alo, ahi, B0, nb_end, img, lo, hi, t_cur, Bp, alo_p, ahi_p, wv =', [x if x < 2**63 else x - 2**64 for x in hh[40:55]])
    names = ['control', 'ldsread', 'rows', 'tail', 'barrier', 'readout', 'ld issue', 'st issue', 'wait+bar', '-', '-', 'n_active', 'total']
    out = (C.c_ulonglong * 256)()
    lib.lqrhip_band_timing.argtypes = [C.POINTER(C.c_ulonglong)]
    assert lib.lqrhip_band_timing(out) == 0
    a = np.array(out[:], dtype=np.int64).reshape(16, 16)
    print('wave ' + ' '.join('%10s' % n for n in names))
    for w in range(16):
        if a[w, 12]: print('%4d ' % w + ' '.join('%10d' % a[w, i] for i in range(13)))
    sys.exit(0)
out = (C.c_ulonglong * 256)()
lib.lqrhip_band_timing.argtypes = [C.POINTER(C.c_ulonglong)]
assert lib.lqrhip_band_timing(out) == 0
a = np.array(out[:], dtype=np.int64).reshape(16, 16)
names = ["setup", "top->landed", "landed(c)", "pre-rows", "rows", "handover", "landed(i)", "barrier", "post-bar", "flush", "rebase+issue", "n_active", "total"]
print("band kernel", bk, "images", nimg, "(last launch, image 0; cycles)")
print("wave " + " ".join("%12s" % n for n in names))
for w in range(16):
    if a[w, 12] == 0: continue
    print("%4d " % w + " ".join("%12d" % a[w, i] for i in range(13)))
