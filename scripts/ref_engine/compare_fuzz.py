"""oracle vs the genuine engine on seeded randomised cases (tests/fuzz_cases.py); build container only
usage: compare_fuzz.py [first_seed] [count] [cw] [small_only]"""
import os, sys, time, traceback
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, HERE)
import numpy as np
import harness as H, lqr_ctypes as L, fuzz_cases as F
import ref_engine as R

first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cw = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0x37f
small_only = len(sys.argv) > 4 and sys.argv[4] == "1"
variant = os.environ.get("ORACLE_VARIANT")          # x87_64 / x87_53: `make -C oracle x87`
orc = L.Api(os.path.join(ROOT, "oracle", "liblqr_oracle_%s.so" % variant), "o") if variant else L.oracle_api()
same = differ = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    img, nw, nh, kw, what = F.draw_case(rng, small=(small_only or (seed - 5000) % 3 != 0))
    api = R.RefApi(cw)
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
print("seeds %d..%d cw=%#x: same %d, differ %d, %.0f s" % (first, first + count - 1, cw, same, differ, time.time() - t0))
