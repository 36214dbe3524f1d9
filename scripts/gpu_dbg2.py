import sys; sys.path.insert(0,"tests")
import numpy as np, lqr_ctypes as L, datasets as D, harness as H
o=L.oracle_api(); e=L.engine_api()
for (w,h,s,seed,freq) in [(3840,2160,60,100,0),(3840,2160,60,100,2)]:
    img=D.noise(w,h,seed)
    a=H.run_case(o,img,w-s,h,switch_freq=freq); b=H.run_case(e,img,w-s,h,switch_freq=freq)
    va,vb=a["vmap"]["data"],b["vmap"]["data"]
    bad=np.argwhere(va!=vb)
    if len(bad):
        lv=np.maximum(va[va!=vb], vb[va!=vb])
        k=lv.min()
        rows=np.unique(bad[lv==k][:,0])
        print(w,h,s,freq,"ndiff",len(bad),"first differing level", k, "rows", rows[:6], "..", rows[-3:], "n rows", len(rows), flush=True)
        y=rows[0]; print("  row", y, "oracle col", np.nonzero(va[y]==k)[0], "engine col", np.nonzero(vb[y]==k)[0])
    else: print(w,h,s,freq,"identical", flush=True)
