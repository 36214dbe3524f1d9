import sys; sys.path.insert(0,"tests")
import numpy as np, lqr_ctypes as L, datasets as D, harness as H
o=L.oracle_api(); e=L.engine_api()
w,h,seed=3840,2160,100
img=D.noise(w,h,seed)
for s in (8, 20, 30, 36):
    snaps=[]
    for api in (o,e):
        api.lqrx_set_debug(1)
        c=L.Carver(api,img).configure(switch_freq=0)
        assert c.resize(w-s,h)==1
        snaps.append(c.debug_snapshot()); api.lqrx_set_debug(0); c.destroy()
    (ea,ma,da),(eb,mb,db)=snaps
    bm=np.argwhere(ma!=mb); bd=np.argwhere(da[1:]!=db[1:])
    print("seams",s,"m diffs",len(bm),"least diffs",len(bd), "en diffs", (ea!=eb).sum(), flush=True)
    if len(bm): 
        y,x=bm[0]; print("  first m diff at",y,x,ma[y,x],mb[y,x], "rows with diffs", np.unique(bm[:,0])[:8])
    if len(bd):
        y,x=bd[0]; print("  first least diff at row",y+1,"col",x,da[y+1,x],db[y+1,x], "rows", np.unique(bd[:,0]+1)[:8], "cols", np.unique(bd[:,1])[:8])
