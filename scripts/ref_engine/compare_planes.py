"""Plane-level comparison of the oracle with the genuine engine's own memory (build container only): rigidity table,
bias and rigidity-mask planes, energies for every built-in function x channel layout (with and without bias), and the DP
planes m / back pointer after the full build and after k incremental updates.
usage: compare_planes.py [mode]      mode: sse (default) | 0x37f | 0x27f"""
import atexit, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, HERE)
import numpy as np
import datasets as D, harness as H, lqr_ctypes as L
import ref_engine as R, stepper as S

mode = sys.argv[1] if len(sys.argv) > 1 else "sse"
cw, f24 = (0x27f, True) if mode == "sse" else (int(mode, 0), False)
orc = L.oracle_api()
apis = []


def ulp(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia); ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    d = np.abs(ia - ib)
    return int((d != 0).sum()), int(d.max()) if d.size else 0


def pair(img, nw, nh, **kw):
    api = R.RefApi(cw, float24=f24)
    apis.append(api)
    c, _ = H.init_carver(api, img, nw, nh, **kw)
    o, _ = H.init_carver(orc, img, nw, nh, **kw)
    return c, o


tot = dict(checks=0, bad=0)


def report(what, n, mx, size):
    tot["checks"] += 1
    tot["bad"] += n != 0
    print("%-78s %s" % (what, "identical (%d values)" % size if n == 0 else "DIFFERENT: %d of %d values, max %d ulp" % (n, size, mx)), flush=True)


# 1. rigidity table after init (lqr_carver.c:254-256 of liblqr 0.4.1; genuine 0x4107a4-0x4107e2)
for rig in (0.5, 1.0, 3.0, 10.0, 33.3, 100.0):
    for delta in (1, 2, 5, 16):
        for h in (3, 37, 100, 1080):
            img = D.noise(8, h, 1, channels=1)
            c, o = pair(img, 7, h, rigidity=rig, delta_x=delta)
            st = S.Stepper(c)
            tab = st.table("rigidity_map", 2 * delta + 1, -delta)
            exp = np.array([np.float32(np.float32(np.float32(rig) * np.float32(abs(x)) ** np.float32(1.5)) / np.float32(h)) for x in range(-delta, delta + 1)], np.float32)
            # the oracle's own table through its DP is checked below; here the formula as the oracle states it (powf, float ops)
            import ctypes
            libm = ctypes.CDLL("libm.so.6"); libm.powf.restype = ctypes.c_float; libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
            exp = np.array([np.float32(np.float32(np.float32(rig) * np.float32(libm.powf(abs(x), 1.5))) / np.float32(h)) for x in range(-delta, delta + 1)], np.float32)
            n, mx = ulp(tab, exp)
            if n:
                report("rigidity table rig %g delta %d h %d" % (rig, delta, h), n, mx, tab.size)
            tot["checks"] += 1; tot["bad"] += n != 0
            c.destroy(); o.destroy(); apis.pop().close()
print("rigidity tables: %d checked, %d differ" % (tot["checks"], tot["bad"]), flush=True)

# 2. energies, all functions x channel layouts, without and with bias (E3/E4; bias/w_start added in compute_e)
for ch in (1, 2, 3, 4):
    for nrg in range(7):
        for masks in (False, True):
            w, h = 97, 61
            img = (D.alpha_ramp if ch in (2, 4) else D.photo_like)(w, h, 40 + ch, channels=ch)
            kw = dict(nrg_func=nrg)
            if masks:
                pres = D.photo_like(w, h, 99, channels=4)          # a mask with all sorts of values
                kw.update(pres=pres, pres_coeff=777, disc=D.alpha_ramp(w, h, 98, channels=2), disc_coeff=313)
            c, o = pair(img, w - 1, h, **kw)
            st = S.Stepper(c)
            st.begin(2)
            en, m, dx = st.planes()
            eo = o.energy()
            n, mx = ulp(en, eo)
            report("energy ch %d nrg %d %s" % (ch, nrg, "bias" if masks else "    "), n, mx, en.size)
            if masks:
                nb = w * h
                b = st.arr("bias", np.float32, nb)
            c.destroy(); o.destroy(); apis.pop().close()

# 3. DP planes after the full build and after k incremental updates (E5, E9), delta / rigidity / rigidity-mask variants
orc.lqrx_set_debug(1)
for name, kw in (("plain", {}), ("delta2", dict(delta_x=2)), ("delta5", dict(delta_x=5)), ("rigidity", dict(rigidity=7.0)),
                 ("rigidity delta2", dict(rigidity=3.0, delta_x=2)), ("rigidity delta4", dict(rigidity=3.0, delta_x=4)),
                 ("rigmask", dict(rigidity=2.0, rigmask=D.photo_like(301, 157, 5, channels=4))),
                 ("masks", dict(pres=D.ellipse_mask(301, 157), disc=D.band_mask(301, 157, 40, 90))),
                 ("null energy + masks", dict(nrg_func=6, pres=D.ellipse_mask(301, 157), disc=D.band_mask(301, 157, 40, 90)))):
    for gen in (D.photo_like, D.noise, D.flat_blocks):
        w, h, k = 301, 157, 40
        img = gen(w, h, 17)
        kw2 = dict(kw); kw2["switch_freq"] = 0
        c, o = pair(img, w - k, h, **kw2)
        st = S.Stepper(c)
        st.begin(k + 1)
        for i in range(k):
            st.seam()
        en, m, dx = st.planes()
        assert o.resize(w - k, h) == 1
        eo, mo, do = o.debug_snapshot()
        (n1, x1), (n2, x2) = ulp(en, eo), ulp(m, mo)
        nd = int((dx[1:] != do[1:]).sum())
        stale = int((dx == -999).sum())
        report("after %d incremental updates: %s, %s: en" % (k, name, gen.__name__), n1, x1, en.size)
        report("after %d incremental updates: %s, %s: m" % (k, name, gen.__name__), n2, x2, m.size)
        report("after %d incremental updates: %s, %s: back pointers (%d stale in the genuine plane)" % (k, name, gen.__name__, stale), nd, 0, dx.size)
        c.destroy(); o.destroy(); apis.pop().close()
orc.lqrx_set_debug(0)
print("mode %s: %d checks, %d with differences" % (mode, tot["checks"], tot["bad"]))
