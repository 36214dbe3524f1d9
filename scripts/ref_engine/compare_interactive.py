"""oracle vs the genuine engine on the INTERACTIVE path (render_interactive, render.c:465-574): one persistent carver per
library, a random sequence of resizes inside and beyond the cached map and flattens; after every call the return value,
getters, image and dumped map must agree (the sequences of scripts/fuzz_interactive.py).  Build container only.
usage: compare_interactive.py [first_seed] [count] [cw]"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, HERE)
import numpy as np
import harness as H, lqr_ctypes as L, datasets as D
import ref_engine as R
import ref_cases as C


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    cw = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0x1027f
    orc = L.oracle_api()
    same = differ = calls = 0
    t0 = time.time()
    for seed in range(first, first + count):
        img, kw, mk, steps, what = C.interactive_case(np.random.default_rng(seed))
        h, w = img.shape[:2]
        api = R.RefApi(cw & 0xffff, float24=bool(cw & 0x10000))
        try:
            s0 = steps[0]
            cs = [H.init_carver(a, img, s0[1] if s0[0] == "r" else w, s0[2] if s0[0] == "r" else h, **kw, **mk)[0] for a in (api, orc)]
            for st in steps:
                rets = [c.resize(st[1], st[2]) if st[0] == "r" else c.flatten() for c in cs]
                assert rets[0] == rets[1], "return values %s at %s" % (rets, st)
                if rets[0] != L.LQR_OK:
                    break
                calls += 1
                assert cs[0].getters() == cs[1].getters(), "getters at %s: %s %s" % (st, cs[0].getters(), cs[1].getters())
                assert np.array_equal(cs[0].read_image(), cs[1].read_image()), "image at %s" % (st,)
                va, vb = cs[0].vmap_dump(), cs[1].vmap_dump()
                assert va["depth"] == vb["depth"] and np.array_equal(va["data"], vb["data"]), "map at %s" % (st,)
            cs[1].destroy()
            same += 1
        except (AssertionError, R.RefCrash) as e:
            differ += 1
            print("DIFFER", seed, what, "|", str(e)[:200], flush=True)
        api.close()
    print("interactive seeds %d..%d cw=%#x: same %d (%d calls compared), differ %d, %.0f s" % (first, first + count - 1, cw, same, calls, differ, time.time() - t0))
