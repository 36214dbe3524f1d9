import sys, time
sys.path.insert(0, 'tests')
import numpy as np
import lqr_ctypes as L, datasets as D, harness as H
o = L.oracle_api(); e = L.engine_api()
img = D.photo_like(96, 64, 1)
print(img[0,:8])
for ef in (2,):
    cb = L.Carver(e, img).configure(nrg_func=ef)
    eb = cb.energy()
    print(eb[0,:8]); print(eb[1,:8]); print(eb[63,:8])
    cb2 = L.Carver(e, img); cb2.configure(nrg_func=1); cb2.configure(nrg_func=2)
    print(cb2.energy()[0,:8])
    cb3 = L.Carver(e, img); 
    print(cb3.energy()[0,:8])
