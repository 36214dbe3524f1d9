"""oracle vs the genuine engine on seeded randomised cases (tests/fuzz_cases.py); build container only
usage: compare_fuzz.py [first_seed] [count] [cw] [small_only]"""
import os, sys, time, traceback
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, HERE)
import numpy as np
import harness as H, lqr_ctypes as L, fuzz_cases as F, datasets as D
import ref_engine as R

first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cw = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0x37f
small_only = len(sys.argv) > 4 and sys.argv[4] == "1"
variant = os.environ.get("ORACLE_VARIANT")          # x87_64 / x87_53: `make -C oracle x87`
orc = L.Api(os.path.join(ROOT, "oracle", "liblqr_oracle_%s.so" % variant), "o") if variant else L.oracle_api()
same = differ = defect = 0


def map_is_valid(v):
    """every line across the seams carries every level 1..depth exactly once"""
    d = v["data"] if v["orientation"] == 0 else v["data"].T
    want = np.arange(1, v["depth"] + 1)
    for row in d:
        lv = np.sort(row[row != 0])
        if lv.size != want.size or not np.array_equal(lv, want):
            return False
    return True

t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    img, nw, nh, kw, what = F.draw_case(rng, small=(small_only or (seed - 5000) % 3 != 0))
    if os.environ.get("FUZZ_EXTRAS"):      # the plug-in's other switches (scripts/fuzz_parity.py's second stream)
        extras = np.random.default_rng(seed + 1000003)
        ex = dict(output_seams=bool(extras.random() < 0.3), resize_aux_layers=bool(extras.random() < 0.3), scaleback=bool(extras.random() < 0.25),
                  no_disc_on_enlarge=bool(extras.random() < 0.7), enl_step=float(extras.choice([150.0, 110.0, 200.0])))
        if "pres" not in kw and extras.random() < 0.5:
            h_, w_ = img.shape[:2]
            kw.update(pres=D.ellipse_mask(w_, h_), disc=D.band_mask(w_, h_, w_ // 5, w_ // 3))
        kw.update(ex)
        what += " extras:%s" % {k: v for k, v in ex.items() if v not in (False, 150.0)}
    api = R.RefApi(cw & 0xffff, float24=bool(cw & 0x10000))          # cw 0x1027f = 'sse' mode
    try:
        a = H.run_case(api, img, nw, nh, progress=True, **kw)
        hc = api.r.heap_check()
        b = H.run_case(orc, img, nw, nh, progress=True, **kw)
        try:
            H.assert_same(a, b, what)
            same += 1
            if hc["bad"] or hc["freed_bad"]:
                print("SAME but heap", seed, what, hc)
        except AssertionError as e:
            maps = [a["vmap"]] + a.get("vmaps", [])
            if "vmaps" not in a:          # a map of the first direction is flattened away: run again keeping every map
                api2 = R.RefApi(cw & 0xffff, float24=bool(cw & 0x10000))
                try:
                    maps += H.run_case(api2, img, nw, nh, **dict(kw, output_seams=True))["vmaps"]
                    hc2 = api2.r.heap_check()
                    hc["bad"] += hc2["bad"]; hc["freed_bad"] += hc2["freed_bad"]
                except R.RefCrash:
                    hc["bad"] += 1
                api2.close()
            if not all(map_is_valid(v) for v in maps) or hc["bad"] or hc["freed_bad"]:
                defect += 1
                print("GENUINE-DEFECT", seed, what, "| the genuine engine's seam map is not a valid map (a level twice / missing in a line)"
                      " or it wrote past a heap block: heap", hc["bad"], hc["freed_bad"], "| oracle map valid:",
                      all(map_is_valid(v) for v in [b["vmap"]] + b.get("vmaps", [])), flush=True)
            else:
                differ += 1
                print("DIFFER", seed, str(e)[:300], "heap", hc["bad"], hc["freed_bad"], flush=True)
    except R.RefCrash as e:
        differ += 1
        print("CRASH ", seed, what, e, flush=True)
    except Exception as e:
        differ += 1
        traceback.print_exc()
        print("ERROR ", seed, what, repr(e)[:200], flush=True)
    api.close()
print("seeds %d..%d cw=%#x: same %d, genuine-defect %d, differ %d, %.0f s" % (first, first + count - 1, cw, same, defect, differ, time.time() - t0))
