"""Golden vectors made by EXECUTING the reference author's own liblqr 0.4.1 build -- the engine statically linked into
gimp-lqr-plugin.exe inside /root/reference/windows_installer_files/lqr-pack4win/.zip (winpack.sh:8,52-57) -- on inputs
generated here (scripts/ref_engine/: a sandboxed i386 loader; build container only; only the DATA under tests/golden/ref/
travels).  This is what pins the oracle: `-m "not gpu"` checks the CPU oracle against every vector, `-m gpu` checks the HIP
engine against every vector directly.

Evaluation mode of the vectors: "sse" -- the genuine x87 code run with double arithmetic at 53 bits and its float-only
functions at 24 bits, i.e. as an x86-64 / SSE2 build of the same source evaluates.  MANIFEST.json
records per vector whether the build AS SHIPPED (x87 control word 0x37f) gives the same result, and `genuine_defect` for
the inputs on which the genuine engine itself produces an invalid seam map (a level twice / missing in a line, or a write
past a heap block): those are compared on validity only (spec delta 6).
"""
import hashlib
import json
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import harness as H
import lqr_ctypes as L
import ref_cases as C

REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref")
MANIFEST = json.load(open(os.path.join(REF, "MANIFEST.json")))
GETTERS = ("width", "height", "channels", "ref_width", "ref_height", "orientation", "depth")


def vectors(*groups):
    return [v for v in MANIFEST["vectors"] if v["group"] in groups]


def ids(vs):
    return ["%s-%s" % (v["group"], v["name"]) for v in vs]


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def map_is_valid(data, depth, orientation):
    """every line across the seams carries every level 1..depth exactly once (1..depth+1 when the map goes down to width 1:
    finish_vsmap gives the last column a level too)"""
    d = data if orientation == 0 else data.T
    for row in d:
        lv = np.sort(row[row != 0])
        if not (lv.size in (depth, depth + 1) and np.array_equal(lv, np.arange(1, lv.size + 1)) and (lv.size == depth or lv.size == row.size)):
            return False
    return True


def load_input(entry, z):
    spec = entry["spec"]
    if spec["kind"] == "fixture":
        kw = {}
        for k in z.files:
            if k.startswith("kw_"):
                v = z[k]
                kw[k[3:]] = v if v.ndim else v.item()
        img = z["img"]
        nw, nh = [int(v) for v in z["new_size"]]
    elif spec["kind"] in ("fuzz", "extras"):
        img, nw, nh, kw, _ = C.fuzz_input(spec)
    else:
        img, nw, nh, kw, _ = C.config_input(spec["name"])
    assert sha(img) == str(z["input_sha1"]), "the input generator does not reproduce the input the vector was made from"
    assert [nw, nh] == [int(v) for v in z["new_size"]]
    return img, nw, nh, kw


def same_map(data, z, key, what):
    """big maps are stored as SHA-1 only"""
    assert sha(data) == str(z[key + "_sha1"]), what
    if key in z.files:
        assert np.array_equal(data, z[key]), what


def compare(r, z, what):
    assert [r["ret"], r.get("ret2", -1)] == list(z["rets"]), what
    assert [r["getters"][k] for k in GETTERS] == list(z["getters"]), what
    assert np.float32(r["getters"]["enl_step"]) == z["enl_step"], what
    assert [r["vmap"]["depth"], r["vmap"]["orientation"]] == list(z["vmap_meta"]), what
    same_map(r["vmap"]["data"], z, "vmap", what + ": seam indices differ from the genuine engine's")
    assert sha(r["image"]) == str(z["image_sha1"]), what + ": pixels differ from the genuine engine's"
    if "image" in z.files:
        assert np.array_equal(r["image"], z["image"]), what
    for i, a in enumerate(r["aux"]):
        assert sha(a) == str(z["aux%d_sha1" % i]), what + ": attached layer %d" % i
    dumped = r.get("vmaps", [])
    assert len(dumped) == len([k for k in z.files if k.startswith("dumped") and k.endswith("_meta")]), what
    for i, v in enumerate(dumped):
        assert [v["depth"], v["orientation"]] == list(z["dumped%d_meta" % i]), what
        same_map(v["data"], z, "dumped%d" % i, what + ": dumped seam map %d" % i)
    assert [list(e) for e in r["events"]] == json.loads(str(z["events"])), what + ": progress events"


def check(api, entry):
    z = np.load(os.path.join(REF, entry["file"]))
    img, nw, nh, kw = load_input(entry, z)
    r = H.run_case(api, img, nw, nh, progress=True, **kw)
    what = "%s %s %s" % (entry["group"], entry["name"], entry.get("what", ""))
    if entry["genuine_defect"]:
        # the genuine engine's own result here is not a valid seam map; ours must be one, and the vector must show the defect
        maps = [(z["vmap"], *z["vmap_meta"])] + [(z[k[:-5]], *z[k]) for k in z.files if k.startswith("dumped") and k.endswith("_meta")]
        assert not all(map_is_valid(d, int(dep), int(o)) for d, dep, o in maps) or any(entry["heap"]), what
        for v in [r["vmap"]] + r.get("vmaps", []):
            assert map_is_valid(v["data"], v["depth"], v["orientation"]), what
        return
    compare(r, z, what)


SMALL = vectors("fixtures", "fuzz", "extras")
CONFIGS = vectors("configs")
CONFIG4 = vectors("config4")
DELTA_WIDE = vectors("deltawide")           # round 6: delta_x 5 .. 10, plain and with rigidity + a rigidity mask + a preservation mask
CONFIG5_HALF = vectors("config5half")       # round 6: config 5 at HALF scale (3840 x 2160, 500 seams, masks, rigidity 10) and its delta_x 2 / rigidity-mask variants


def test_manifest_is_complete():
    assert len(vectors("fixtures")) == 17 and len(vectors("fuzz")) == 60 and len(vectors("extras")) == 40
    assert len(CONFIG4) == 64 and len(CONFIGS) >= 5 and len(vectors("interactive")) == 40 and len(CONFIG5_HALF) == 3 and len(DELTA_WIDE) == 12
    assert len(MANIFEST["exe_sha256"]) == 64
    for v in MANIFEST["vectors"]:
        assert os.path.exists(os.path.join(REF, v["file"])), v["file"]
    # the inputs on which the genuine engine is defective are the heavily tied ones (NULL energy), nothing else
    for v in SMALL:
        if v["genuine_defect"] and v["spec"]["kind"] != "fixture":
            assert "'nrg_func': 6" in v["what"], v


def test_precision_evidence_covers_every_vector_the_shipped_build_differs_on():
    """PRECISION.json (scripts/ref_engine/precision_evidence.py): for each vector whose result the exe as shipped does not give, which
    single float-only function under the 24-bit word reproduces it on top of 53-bit doubles -- the per-function evidence behind the "sse" mode"""
    prec = json.load(open(os.path.join(REF, "PRECISION.json")))
    have = {(v["group"], v["name"]) for v in prec["vectors"]}
    want = {(v["group"], v["name"]) for v in MANIFEST["vectors"] if v.get("same_as_shipped") is False and v["group"] != "interactive"}
    assert want == have
    for v in prec["vectors"]:
        r = v["reproduces_the_vector"]
        assert r["sse"] and not r["shipped"], v["name"]
        # nothing but the two DP functions (or the double arithmetic alone) ever decides a result
        assert r["d53"] or set(v["changes_d53_result"]) <= {"lqr_carver_build_mmap", "lqr_carver_update_mmap"}, v


@pytest.mark.parametrize("entry", SMALL, ids=ids(SMALL))
def test_oracle_reproduces_the_genuine_engine(oracle, entry):
    check(oracle, entry)


@pytest.mark.parametrize("entry", CONFIGS, ids=ids(CONFIGS))
def test_oracle_reproduces_the_genuine_engine_on_baseline_configs(oracle, entry):
    """BASELINE.json's configs 1 and 2 at full size, 3 at full size (both directions, both seam maps), 5 and its variants at
    quarter scale"""
    check(oracle, entry)


@pytest.mark.parametrize("entry", DELTA_WIDE, ids=ids(DELTA_WIDE))
def test_oracle_reproduces_the_genuine_engine_with_wide_delta(oracle, entry):
    check(oracle, entry)


def test_oracle_reproduces_the_genuine_engine_on_config5_at_half_scale(oracle):
    """config 5 at the largest size the 32-bit runner's arena holds (config 4's images are 3840 x 2160 too): 500 seams of a 4K image with
    the preservation ellipse, the discard band and rigidity 10; delta_x 2; a rigidity mask (~50 s of the oracle each, side by side)"""
    with ThreadPoolExecutor(3) as ex:
        list(ex.map(lambda e: check(oracle, e), CONFIG5_HALF))


def test_oracle_reproduces_the_genuine_engine_on_config4_images(oracle):
    """four of config 4's 64 4K images on the CPU (the GPU run compares all 64)"""
    with ThreadPoolExecutor(4) as ex:
        list(ex.map(lambda e: check(oracle, e), CONFIG4[::21]))


def interactive_session(api, entry):
    img, kw, mk, steps, what = C.interactive_case(np.random.default_rng(entry["seed"]))
    h, w = img.shape[:2]
    s0 = steps[0]
    c = H.init_carver(api, img, s0[1] if s0[0] == "r" else w, s0[2] if s0[0] == "r" else h, **kw, **mk)[0]
    for st, exp in zip(steps, entry["steps"]):
        ret = c.resize(st[1], st[2]) if st[0] == "r" else c.flatten()
        assert ret == exp["ret"], (what, st)
        if ret != L.LQR_OK:
            break
        v = c.vmap_dump()
        if entry["genuine_defect"]:
            assert map_is_valid(v["data"], v["depth"], v["orientation"])
            continue
        g = c.getters()
        assert {k: g[k] for k in GETTERS} == exp["getters"], (what, st)
        assert [v["depth"], v["orientation"]] == exp["vmap_meta"] and sha(v["data"]) == exp["vmap_sha1"], (what, st, "seam map")
        assert sha(c.read_image()) == exp["image_sha1"], (what, st, "pixels")
    c.destroy()


INTERACTIVE = json.load(open(os.path.join(REF, "interactive.json"))) if os.path.exists(os.path.join(REF, "interactive.json")) else []


@pytest.mark.parametrize("entry", INTERACTIVE, ids=[e["name"] for e in INTERACTIVE])
def test_oracle_reproduces_the_genuine_engine_interactive(oracle, entry):
    """render_interactive's persistent carver (render.c:465-574): resizes inside and beyond the cached map, flattens"""
    interactive_session(oracle, entry)


def planes_index():
    z = np.load(os.path.join(REF, "planes.npz"))
    return z, json.loads(str(z["index"]))


def check_planes(api):
    z, index = planes_index()
    api.lqrx_set_debug(1)
    try:
        for e in index:
            if e["kind"] == "energy":
                img, nw, nh, kw = C.energy_case(e["ch"], e["nrg"], e["masks"])
                c, _ = H.init_carver(api, img, nw, nh, **kw)
                en = c.energy()
                c.destroy()
                assert np.array_equal(en.view(np.int32), z[e["key"]].view(np.int32)), "energy plane %s: not bit-identical (0 ULP) to the genuine engine's" % e["key"]
            else:
                img, nw, nh, kw = C.dp_case(e["variant"])
                c, _ = H.init_carver(api, img, nw, nh, **kw)
                assert c.resize(nw, nh) == L.LQR_OK
                en, m, dx = c.debug_snapshot()
                c.destroy()
                k = e["key"]
                assert np.array_equal(en.view(np.int32), z[k + "_en"].view(np.int32)), k + ": energies after the incremental updates"
                assert np.array_equal(m.view(np.int32), z[k + "_m"].view(np.int32)), k + ": cumulative minima (m) after the incremental updates"
                assert np.array_equal(dx[1:], z[k + "_dx"][1:].astype(np.int32)), k + ": back pointers after the incremental updates"
    finally:
        api.lqrx_set_debug(0)


def test_oracle_planes_equal_the_genuine_engines_memory(oracle):
    """energies of all 7 built-in functions x 4 channel layouts (with and without bias) and the DP planes (m, back pointers)
    after 40 incremental updates, bit for bit against planes read out of the genuine engine's memory"""
    check_planes(oracle)


# ---------------------------------------------------------------- the HIP engine against the genuine engine's vectors
@pytest.mark.gpu
@pytest.mark.parametrize("entry", SMALL, ids=ids(SMALL))
def test_engine_reproduces_the_genuine_engine(engine, entry):
    check(engine, entry)


@pytest.mark.gpu
@pytest.mark.parametrize("entry", CONFIGS, ids=ids(CONFIGS))
def test_engine_reproduces_the_genuine_engine_on_baseline_configs(engine, entry):
    check(engine, entry)


@pytest.mark.gpu
@pytest.mark.parametrize("entry", DELTA_WIDE, ids=ids(DELTA_WIDE))
def test_engine_reproduces_the_genuine_engine_with_wide_delta(engine, entry):
    """delta_x 5 .. 10 on k_dp_tile_p's general instantiations (round 6) against what the genuine liblqr produced"""
    check(engine, entry)


@pytest.mark.gpu
@pytest.mark.parametrize("entry", CONFIG5_HALF, ids=ids(CONFIG5_HALF))
def test_engine_reproduces_the_genuine_engine_on_config5_at_half_scale(engine, entry):
    check(engine, entry)


@pytest.mark.gpu
@pytest.mark.parametrize("entry", INTERACTIVE, ids=[e["name"] for e in INTERACTIVE])
def test_engine_reproduces_the_genuine_engine_interactive(engine, entry):
    interactive_session(engine, entry)


@pytest.mark.gpu
def test_engine_planes_equal_the_genuine_engines_memory(engine):
    check_planes(engine)


@pytest.mark.gpu
def test_engine_config4_all_64_images_equal_the_genuine_engine(engine):
    """BASELINE config 4 as stated, one lock-step batch of 64 4K images: EVERY image's seam map and pixels against what the
    genuine liblqr produced for it"""
    import datasets as D
    imgs = [C.config_input(e["spec"]["name"])[0] for e in CONFIG4]
    cs = [L.Carver(engine, im).configure() for im in imgs]
    assert L.resize_batch(engine, cs, 3640, 2160) == L.LQR_OK
    for c, e in zip(cs, CONFIG4):
        z = np.load(os.path.join(REF, e["file"]))
        v = c.vmap_dump()
        assert [v["depth"], v["orientation"]] == list(z["vmap_meta"]), e["name"]
        same_map(v["data"], z, "vmap", e["name"] + ": seam indices")
        assert sha(c.read_image()) == str(z["image_sha1"]), e["name"] + ": pixels"
    for c in cs:
        c.destroy()
