"""Per-function evidence for the "sse" evaluation mode of the golden vectors.  BUILD CONTAINER ONLY.

The vectors are made with the genuine x87 code under control word 0x27f (53-bit mantissa: double arithmetic as an SSE2 build performs
it) and its seven float-only functions under 0x07f (24 bits: float arithmetic as SSE2 performs it).  For every vector on which the
exe AS SHIPPED (0x37f everywhere) gives another result, this script runs the input under
  * "shipped"  0x37f, nothing spliced
  * "d53"      0x27f, nothing spliced            (only the double arithmetic changed)
  * "d53+<f>"  0x27f and 0x07f inside ONE function f of the seven
  * "sse"      0x27f and 0x07f inside all seven  (the vectors' mode)
and records which of them reproduce the vector: which function's precision word the result really hangs on.
    python scripts/ref_engine/precision_evidence.py            -> tests/golden/ref/PRECISION.json"""
import json, os, sys
from concurrent.futures import ProcessPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, HERE)
import numpy as np
import harness as H
import make_ref_golden as G
import ref_engine as R

OUT = os.path.join(ROOT, "tests", "golden", "ref")


def run_mode(mode, img, nw, nh, kw):
    if mode == "shipped":
        a = R.RefApi(0x37f)
    elif mode == "d53":
        a = R.RefApi(0x27f)
    elif mode == "sse":
        a = R.RefApi(0x27f, float24=True)
    else:
        a = R.RefApi(0x27f, float24_only=[mode])
    try:
        return G.digest(H.run_case(a, img, nw, nh, progress=True, **kw))
    finally:
        a.close()


def one(entry):
    img, nw, nh, kw, _ = G.make_input(dict(entry["spec"]))
    modes = ["shipped", "d53"] + list(R.FLOAT_ONLY) + ["sse"]
    dig = {m: run_mode(m, img, nw, nh, kw) for m in modes}
    ref = dig["sse"]
    return dict(group=entry["group"], name=entry["name"],
                reproduces_the_vector={m: dig[m] == ref for m in modes},
                decisive=[m for m in R.FLOAT_ONLY if dig[m] == ref],
                changes_d53_result=[m for m in R.FLOAT_ONLY if dig[m] != dig["d53"]])


def main():
    man = json.load(open(os.path.join(OUT, "MANIFEST.json")))
    todo = [v for v in man["vectors"] if v.get("same_as_shipped") is False and v["group"] != "interactive"]
    res = []
    with ProcessPoolExecutor(max_workers=int(os.environ.get("JOBS", "6"))) as ex:
        for r in ex.map(one, todo):
            print(r["group"], r["name"], "decisive alone:", r["decisive"], "| change the 53-bit result:", r["changes_d53_result"], "| d53 alone reproduces:", r["reproduces_the_vector"]["d53"], flush=True)
            res.append(r)
    json.dump(dict(note="which single float-only function, run under the 24-bit control word on top of 53-bit doubles, reproduces the vector (made with all seven): "
                        "scripts/ref_engine/precision_evidence.py", functions=list(R.FLOAT_ONLY), vectors=res), open(os.path.join(OUT, "PRECISION.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
