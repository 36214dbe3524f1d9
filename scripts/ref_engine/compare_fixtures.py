"""oracle vs the genuine engine on the cases of tests/golden/make_golden.py (build container only)"""
import os, sys, traceback
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, HERE)
import numpy as np
import harness as H, lqr_ctypes as L
import make_golden as G
import ref_engine as R

cw = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0x37f
variant = os.environ.get("ORACLE_VARIANT")          # x87_64 / x87_53: `make -C oracle x87`
orc = L.Api(os.path.join(ROOT, "oracle", "liblqr_oracle_%s.so" % variant), "o") if variant else L.oracle_api()
for name, (img, nw, nh, kw) in G.cases().items():
    api = R.RefApi(cw & 0xffff, float24=bool(cw & 0x10000))          # cw 0x1027f = 'sse' mode
    try:
        a = H.run_case(api, img, nw, nh, progress=True, **kw)
        hc = api.r.heap_check()
        b = H.run_case(orc, img, nw, nh, progress=True, **kw)
        try:
            H.assert_same(a, b, name)
            print("SAME   ", name, "heap", hc["bad"], hc["freed_bad"])
        except AssertionError as e:
            print("DIFFER ", name, str(e)[:200], "heap", hc)
    except R.RefCrash as e:
        print("CRASH  ", name, e)
    except Exception as e:
        traceback.print_exc()
        print("ERROR  ", name, repr(e)[:200])
    api.close()
