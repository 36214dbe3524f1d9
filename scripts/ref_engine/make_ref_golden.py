"""Golden vectors produced by EXECUTING the reference author's own liblqr build (refrun.c / ref_engine.py) on inputs
generated here.  BUILD CONTAINER ONLY; only the DATA this writes (tests/golden/ref/*.npz + MANIFEST.json) travels.

Evaluation mode of the x87 code: "sse" (control word 0x27f, the float-only DP functions under 0x07f): arithmetic as an
x86-64 / SSE2 build of the same source performs it -- the platform this repository's "bit-exact" refers to.  For every
vector the manifest also records whether the build AS SHIPPED (control word 0x37f) produces the same result.

    python scripts/ref_engine/make_ref_golden.py [group ...]     groups: fixtures fuzz extras interactive planes configs config4 config5half deltawide
"""
import hashlib, json, os, sys, time
from concurrent.futures import ProcessPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, HERE)
import numpy as np
import datasets as D, fuzz_cases as F, harness as H, lqr_ctypes as L
import ref_cases as C
import ref_engine as R, stepper as S

OUT = os.path.join(ROOT, "tests", "golden", "ref")
GETTERS = ("width", "height", "channels", "ref_width", "ref_height", "orientation", "depth")
INLINE_IMAGE_BYTES = 96 * 1024


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def api(mode):
    return R.RefApi(0x27f, float24=True) if mode == "sse" else R.RefApi(0x37f)


def map_is_valid(v):
    """every line across the seams carries every level 1..depth exactly once (1..depth+1 when the map goes down to width 1:
    finish_vsmap gives the last column a level too)"""
    d = v["data"] if v["orientation"] == 0 else v["data"].T
    for row in d:
        lv = np.sort(row[row != 0])
        if not (lv.size in (v["depth"], v["depth"] + 1) and np.array_equal(lv, np.arange(1, lv.size + 1)) and (lv.size == v["depth"] or lv.size == row.size)):
            return False
    return True


def run(mode, img, nw, nh, kw):
    a = api(mode)
    try:
        r = H.run_case(a, img, nw, nh, progress=True, **kw)
        r["heap"] = a.r.heap_check()
    finally:
        a.close()
    return r


def digest(r):
    h = hashlib.sha1()
    for part in [r["vmap"]["data"], r["image"]] + r["aux"] + [v["data"] for v in r.get("vmaps", [])]:
        h.update(np.ascontiguousarray(part).tobytes())
    h.update(json.dumps([r["ret"], r.get("ret2"), r["getters"], r["nlines"]], sort_keys=True).encode())
    return h.hexdigest()


BIG_MAP_BYTES = 4 << 20


def pack(r, img, nw, nh, kw, inline_input, keep_big_maps=False):
    """maps above BIG_MAP_BYTES are stored as SHA-1 only (64 4K maps would be 55 MB) unless keep_big_maps"""
    def put_map(arrays, key, data):
        arrays[key + "_sha1"] = np.array(sha(data))
        if data.nbytes <= BIG_MAP_BYTES or keep_big_maps:
            arrays[key] = data
    arrays = dict(new_size=np.array([nw, nh]), vmap_meta=np.array([r["vmap"]["depth"], r["vmap"]["orientation"]]),
                  getters=np.array([r["getters"][k] for k in GETTERS]), enl_step=np.float32(r["getters"]["enl_step"]),
                  image_sha1=np.array(sha(r["image"])), input_sha1=np.array(sha(img)), rets=np.array([r["ret"], r.get("ret2", -1)]),
                  events=np.array(json.dumps(r["events"])))
    put_map(arrays, "vmap", r["vmap"]["data"])
    if inline_input:
        arrays["img"] = img
        for k, v in kw.items():
            arrays["kw_" + k] = np.asarray(v)
    if r["image"].nbytes <= INLINE_IMAGE_BYTES:
        arrays["image"] = r["image"]
    for i, a in enumerate(r["aux"]):
        arrays["aux%d_sha1" % i] = np.array(sha(a))
        if a.nbytes <= INLINE_IMAGE_BYTES:
            arrays["aux%d" % i] = a
    for i, v in enumerate(r.get("vmaps", [])):
        put_map(arrays, "dumped%d" % i, v["data"])
        arrays["dumped%d_meta" % i] = np.array([v["depth"], v["orientation"]])
    return arrays


def one(task):
    """(group, name, how-to-make-the-input) -> writes the vector, returns its manifest entry"""
    group, name, spec = task
    img, nw, nh, kw, inline = make_input(spec)
    t0 = time.time()
    r = run("sse", img, nw, nh, kw)
    t_sse = time.time() - t0
    shipped = run("shipped", img, nw, nh, kw)
    maps = [r["vmap"]] + r.get("vmaps", [])
    defect = not all(map_is_valid(v) for v in maps) or bool(r["heap"]["bad"] or r["heap"]["freed_bad"])
    entry = dict(group=group, name=name, spec=spec, what=spec.get("what", ""), genuine_defect=defect, heap=[r["heap"]["bad"], r["heap"]["freed_bad"]],
                 same_as_shipped=digest(r) == digest(shipped), shipped_events_same=r["events"] == shipped["events"],
                 seconds=round(t_sse, 2), file="%s_%s.npz" % (group, name))
    np.savez_compressed(os.path.join(OUT, entry["file"]), **pack(r, img, nw, nh, kw, inline, keep_big_maps=name in ("image00", "image63", "config2_fhd")))
    return entry


def make_input(spec):
    k = spec["kind"]
    if k == "fixture":
        import make_golden as G
        img, nw, nh, kw = G.cases()[spec["name"]]
        return img, nw, nh, kw, True
    if k in ("fuzz", "extras"):
        img, nw, nh, kw, what = C.fuzz_input(spec)
        spec["what"] = what
        return img, nw, nh, kw, False
    if k == "config":
        return C.config_input(spec["name"])
    raise ValueError(k)


def interactive(seed):
    img, kw, mk, steps, what = C.interactive_case(np.random.default_rng(seed))
    h, w = img.shape[:2]
    out = []
    for mode in ("sse", "shipped"):
        a = api(mode)
        s0 = steps[0]
        c = H.init_carver(a, img, s0[1] if s0[0] == "r" else w, s0[2] if s0[0] == "r" else h, **kw, **mk)[0]
        rec = []
        valid = True
        for st in steps:
            ret = c.resize(st[1], st[2]) if st[0] == "r" else c.flatten()
            if ret != 1:
                rec.append(dict(ret=ret))
                break
            v = c.vmap_dump()
            valid = valid and map_is_valid(v)
            rec.append(dict(ret=ret, getters={k: c.getters()[k] for k in GETTERS}, image_sha1=sha(c.read_image()), vmap_sha1=sha(v["data"]),
                            vmap_meta=[v["depth"], v["orientation"]]))
        hc = a.r.heap_check()
        a.close()
        out.append((rec, valid and not (hc["bad"] or hc["freed_bad"])))
    return dict(group="interactive", name="seed%d" % seed, seed=seed, what=what, steps=out[0][0], genuine_defect=not out[0][1],
                same_as_shipped=out[0][0] == out[1][0])


def planes():
    """energies (7 functions x 4 channel layouts, with and without bias) and the DP planes after 40 incremental updates,
    read out of the genuine engine's memory"""
    arrays, index = {}, []
    a = api("sse")
    for ch in (1, 2, 3, 4):
        for nrg in range(7):
            for masks in (False, True):
                w, h = 97, 61
                key = "en_ch%d_nrg%d_%s" % (ch, nrg, "bias" if masks else "plain")
                ei, enw, enh, ekw = C.energy_case(ch, nrg, masks)
                c, _ = H.init_carver(a, ei, enw, enh, **ekw)
                st = S.Stepper(c)
                st.begin(2)
                arrays[key] = st.planes()[0]
                index.append(dict(key=key, kind="energy", ch=ch, nrg=nrg, masks=masks))
                c.destroy()
    for name in C.DP_VARIANTS:
        img, nw, nh, kw = C.dp_case(name)
        c, _ = H.init_carver(a, img, nw, nh, **kw)
        st = S.Stepper(c)
        k = img.shape[1] - nw
        st.begin(k + 1)
        for _ in range(k):
            st.seam()
        en, m, dx = st.planes()
        arrays["dp_%s_en" % name], arrays["dp_%s_m" % name], arrays["dp_%s_dx" % name] = en, m, dx.astype(np.int8)
        index.append(dict(key="dp_" + name, kind="dp", variant=name, updates=k, stale=int((dx == -999).sum())))
        c.destroy()
    a.close()
    arrays["index"] = np.array(json.dumps(index))
    np.savez_compressed(os.path.join(OUT, "planes.npz"), **arrays)
    return dict(group="planes", name="planes", file="planes.npz", entries=len(index))


def main():
    groups = set(sys.argv[1:]) or {"fixtures", "fuzz", "extras", "interactive", "planes", "configs", "config4"}
    os.makedirs(OUT, exist_ok=True)
    mpath = os.path.join(OUT, "MANIFEST.json")
    manifest = json.load(open(mpath)) if os.path.exists(mpath) else dict(vectors=[])
    keep = [v for v in manifest["vectors"] if v["group"] not in groups]
    tasks = []
    if "fixtures" in groups:
        import make_golden as G
        tasks += [("fixtures", n, dict(kind="fixture", name=n)) for n in G.cases()]
    if "fuzz" in groups:
        tasks += [("fuzz", "seed%d" % s, dict(kind="fuzz", seed=s, small=(s - 5000) % 3 != 0)) for s in range(5000, 5060)]
    if "extras" in groups:
        tasks += [("extras", "seed%d" % s, dict(kind="extras", seed=s, small=True)) for s in range(8000, 8040)]
    if "configs" in groups:
        tasks += [("configs", n, dict(kind="config", name=n)) for n in ("config1_512", "config2_fhd", "config3_4k_bidir", "config5_quarter",
                                                                         "config5_quarter_delta2", "config5_quarter_rigmask")]
    if "config5half" in groups:      # round 6: config 5 at half scale and its two variants (each ~4 GB-seconds of the genuine engine)
        tasks += [("config5half", n, dict(kind="config", name=n)) for n in ("config5_half", "config5_half_delta2", "config5_half_rigmask")]
    if "deltawide" in groups:        # round 6: delta_x 5 .. 10
        tasks += [("deltawide", "dw_%d_%s" % (d, v), dict(kind="config", name="dw_%d_%s" % (d, v))) for d in (5, 6, 7, 8, 9, 10) for v in ("plain", "rig")]
    if "config4" in groups:
        tasks += [("config4", "image%02d" % i, dict(kind="config", name="config4_%d" % i)) for i in range(64)]
    new = []
    with ProcessPoolExecutor(max_workers=int(os.environ.get("JOBS", "7"))) as ex:
        for e in ex.map(one, tasks):
            print(e["group"], e["name"], "defect" if e["genuine_defect"] else "ok", "shipped-same" if e["same_as_shipped"] else "SHIPPED-DIFFERS", e["seconds"], "s", flush=True)
            new.append(e)
        if "interactive" in groups:
            for e in ex.map(interactive, range(1, 41)):
                new.append(e)
    if "interactive" in groups:
        json.dump([e for e in new if e["group"] == "interactive"], open(os.path.join(OUT, "interactive.json"), "w"), indent=0)
        new = [dict(group="interactive", name=e["name"], file="interactive.json", genuine_defect=e["genuine_defect"], same_as_shipped=e["same_as_shipped"])
               if e["group"] == "interactive" else e for e in new]
    if "planes" in groups:
        new.append(planes())
    exe = R.exe_bytes()
    manifest = dict(source="gimp-lqr-plugin.exe inside /root/reference/windows_installer_files/lqr-pack4win/.zip (liblqr 0.4.1 statically linked, "
                           "winpack.sh:8,52-57), executed by scripts/ref_engine/refrun.c",
                    exe_sha256=hashlib.sha256(exe).hexdigest(), mode="sse: x87 control word 0x27f, float-only DP functions under 0x07f",
                    vectors=keep + new)
    json.dump(manifest, open(mpath, "w"), indent=1)
    print("wrote", len(new), "vectors;", sum(1 for v in new if v.get("genuine_defect")), "genuine defects;",
          sum(1 for v in new if v.get("same_as_shipped") is False), "differ from the build as shipped")


if __name__ == "__main__":
    main()
