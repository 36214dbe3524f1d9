import sys, time
sys.path.insert(0, 'tests')
import numpy as np
import lqr_ctypes as L, datasets as D, harness as H
o = L.oracle_api(); e = L.engine_api()
def chk(name, img, nw, nh, **kw):
    t=time.time(); a = H.run_case(o, img, nw, nh, **kw); t1=time.time()-t
    t=time.time(); b = H.run_case(e, img, nw, nh, **kw); t2=time.time()-t
    try:
        H.assert_same(a, b, name); print("OK  ", name, "cpu %.2fs gpu %.2fs" % (t1, t2), flush=True)
    except AssertionError as ex:
        print("FAIL", name, str(ex)[:300], flush=True)
img = D.photo_like(96, 64, 1)
# energy first
for ef in range(7):
    ca = L.Carver(o, img).configure(nrg_func=ef); cb = L.Carver(e, img).configure(nrg_func=ef)
    ea, eb = ca.energy(), cb.energy()
    print("energy", ef, "max ulp-ish diff", np.abs(ea-eb).max(), "exact", np.array_equal(ea, eb), flush=True)
chk("shrink1", img, 95, 64)
chk("shrink16", img, 80, 64)
chk("shrink16-noise", D.noise(96,64,3), 80, 64)
chk("shrink16-flat", D.flat_blocks(96,64,4), 80, 64)
chk("bidir", img, 80, 50)
chk("vert-first", img, 80, 50, res_order=1)
chk("enlarge", img, 110, 64)
chk("enlarge-multi", img, 170, 64)
chk("rgb3", D.photo_like(70, 40, 5, channels=3), 60, 40)
chk("grey", D.photo_like(70, 40, 5, channels=1), 60, 33)
chk("greya", D.alpha_ramp(70, 40, 5, channels=2), 60, 40)
chk("alpha", D.alpha_ramp(120, 90, 6), 100, 90)
chk("rig-d2", img, 80, 64, rigidity=10.0, delta_x=2)
chk("masks", img, 80, 64, pres=D.ellipse_mask(96,64), disc=D.band_mask(96,64,10,25), rigmask=D.top_half_mask(96,64), rigidity=5.0, resize_aux_layers=True, output_seams=True)
chk("p512", D.noise(512,512,1), 462, 512)
chk("wide", D.photo_like(700, 300, 9), 600, 300)
chk("freq-every", img, 60, 64, switch_freq=100)
