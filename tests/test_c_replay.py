"""C replay harness: tests/c/render_replay.c issues the reference's
render_init_carver / render_noninteractive / write_carver_to_layer / write_all_vmaps call sequence
(src/render.c:211-248,318-376; src/io_functions.c:70-131,155-164,292-314) in C against
include/lqr.h.  It is compiled with -Werror in both typedef modes of the header (built-in GLib-free
typedefs; -DLQR_NO_GLIB_TYPEDEFS + a GLib stand-in), so a prototype that does not take what the
plug-in passes fails the build.  Its output must equal what the Python harness (tests/harness.py)
observes through ctypes on the same library.

CPU suite: linked to the oracle (through oracle/oracle_rename.h).  -m gpu: linked to the engine
(liblqr-hip.so), as the plug-in would be, and additionally compared to the oracle.
"""
import os
import struct
import subprocess

import numpy as np
import pytest

import datasets as D
import harness as H
import lqr_ctypes as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "render_replay.c")
BUILD = os.path.join(ROOT, "tests", "c", "build")
CFLAGS = ["-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include")]


def build(target, glib):
    """target: "oracle" / "engine" (linked at build time) or "dlopen" (no carving library linked: --lib=PATH at run time)"""
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "replay_%s%s" % (target, "_glib" if glib else ""))
    cmd = ["gcc"] + CFLAGS + (["-DLQR_NO_GLIB_TYPEDEFS"] if glib else [])
    if target == "dlopen":
        cmd += ["-DREPLAY_DLOPEN", SRC, "-o", exe, "-ldl", "-lm"]
    elif target == "oracle":
        d = os.path.join(ROOT, "oracle")
        cmd += ["-include", os.path.join(d, "oracle_rename.h"), SRC, "-o", exe, "-L" + d, "-l:liblqr_oracle.so", "-Wl,-rpath," + d, "-lm"]
    else:
        d = os.path.join(ROOT, "gimp-lqr-plugin_amd")
        cmd += [SRC, "-o", exe, "-L" + d, "-l:liblqr-hip.so", "-Wl,-rpath," + d, "-lm"]
    subprocess.check_call(cmd)
    return exe


def write_case(path, img, nw, nh, pres=None, disc=None, rigmask=None, **kw):
    v = dict(H.DEFAULTS); v.update(kw)
    h, w, bpp = img.shape
    masks = [m for m in (pres, disc, rigmask) if m is not None]
    mbpp = masks[0].shape[2] if masks else 0
    hd = [w, h, bpp, nw, nh, v["delta_x"], v["nrg_func"], v["res_order"], int(v["output_seams"]), int(v["resize_aux_layers"]),
          int(v["scaleback"]), int(v["no_disc_on_enlarge"]), v["pres_coeff"], v["disc_coeff"], int(pres is not None),
          int(disc is not None), int(rigmask is not None), mbpp, 0, 0]
    with open(path, "wb") as f:
        f.write(struct.pack("<20i", *hd))
        f.write(struct.pack("<2f", v["rigidity"], v["enl_step"]))
        f.write(np.ascontiguousarray(img).tobytes())
        for m in (pres, disc, rigmask):
            if m is not None:
                f.write(np.ascontiguousarray(m).tobytes())


class Reader:
    def __init__(self, path):
        self.b = open(path, "rb").read()
        self.o = 0

    def i(self):
        v = struct.unpack_from("<i", self.b, self.o)[0]
        self.o += 4
        return v

    def arr(self, dtype, shape):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        a = np.frombuffer(self.b, dtype=dtype, count=int(np.prod(shape)), offset=self.o).reshape(shape)
        self.o += n
        return a


def parse_out(path, n_aux):
    r = Reader(path)
    res = {"ret": r.i()}
    nv = r.i()
    res["vmaps"] = []
    for _ in range(nv):
        w, h, depth = r.i(), r.i(), r.i()
        res["vmaps"].append(dict(depth=depth, data=r.arr(np.int32, (h, w))))
    res["ref_width"], res["ref_height"], res["orientation"], res["depth"] = r.i(), r.i(), r.i(), r.i()
    res["height"], res["channels"] = r.i(), r.i()
    res["nlines"] = r.i()
    w, h, bpp = r.i(), r.i(), r.i()
    res["image"] = r.arr(np.uint8, (h, w, bpp))
    res["aux"] = []
    for _ in range(n_aux):
        r.i()                       # lines of the aux read-out
        w, h, bpp = r.i(), r.i(), r.i()
        res["aux"].append(r.arr(np.uint8, (h, w, bpp)))
    res["n_init"], res["n_update"], res["n_end"] = r.i(), r.i(), r.i()
    assert r.o == len(r.b), "trailing bytes in the replay output"
    return res


def cases():
    w, h = 96, 64
    img = D.photo_like(w, h, 11)
    pres, disc, rig = D.ellipse_mask(w, h), D.band_mask(w, h, 10, 30), D.top_half_mask(w, h)
    yield "defaults", (img, 70, 64), {}
    yield "bidirectional_seams", (img, 80, 50), dict(output_seams=True)
    yield "masks_aux_rigidity", (img, 76, 64), dict(pres=pres, disc=disc, rigmask=rig, rigidity=4.0, delta_x=2, resize_aux_layers=True,
                                                    output_seams=True)
    yield "enlarge_vert_first", (D.noise(60, 40, 3, channels=3), 75, 52), dict(res_order=L.LQR_RES_ORDER_VERT, disc=D.band_mask(60, 40, 5, 15, channels=3))
    yield "lqr_back_gray", (D.photo_like(70, 50, 4, channels=1), 55, 40), dict(scaleback=True, nrg_func=L.LQR_EF_GRAD_SUMABS)


ORACLE_SO = os.path.join(ROOT, "oracle", "liblqr_oracle.so")
ENGINE_SO = os.path.join(ROOT, "gimp-lqr-plugin_amd", "liblqr-hip.so")


def compare(exe, api, tmp_path, name, args, kw, lib_args=()):
    img, nw, nh = args
    case, outp = str(tmp_path / (name + ".case")), str(tmp_path / (name + ".out"))
    write_case(case, img, nw, nh, **kw)
    subprocess.check_call([exe] + list(lib_args) + [case, outp])
    n_aux = sum(1 for k in ("pres", "disc", "rigmask") if kw.get(k) is not None) if kw.get("resize_aux_layers") else 0
    c = parse_out(outp, n_aux)
    p = H.run_case(api, img, nw, nh, progress=True, **kw)
    assert c["ret"] == p["ret"] == L.LQR_OK
    g = p["getters"]
    assert (c["ref_width"], c["ref_height"], c["orientation"], c["depth"], c["height"], c["channels"]) == \
        (g["ref_width"], g["ref_height"], g["orientation"], g["depth"], g["height"], g["channels"])
    assert c["nlines"] == p["nlines"]
    assert np.array_equal(c["image"], p["image"]), name + ": images differ between the C replay and the Python harness"
    assert len(c["aux"]) == len(p["aux"])
    for a, b in zip(c["aux"], p["aux"]):
        assert np.array_equal(a, b)
    assert len(c["vmaps"]) == len(p.get("vmaps", []))
    for a, b in zip(c["vmaps"], p.get("vmaps", [])):
        assert a["depth"] == b["depth"] and np.array_equal(a["data"], b["data"])
    ev = [e[0] for e in p["events"]]
    assert (c["n_init"], c["n_update"], c["n_end"]) == (ev.count("init"), ev.count("update"), ev.count("end"))
    return c


@pytest.mark.parametrize("glib", [False, True], ids=["builtin_typedefs", "glib_typedefs"])
def test_c_replay_against_the_oracle(oracle, tmp_path, glib):
    exe = build("oracle", glib)
    for name, args, kw in cases():
        compare(exe, oracle, tmp_path, name, args, kw)


@pytest.mark.gpu
@pytest.mark.parametrize("glib", [False, True], ids=["builtin_typedefs", "glib_typedefs"])
def test_c_replay_against_the_engine(engine, oracle, tmp_path, glib):
    """the plug-in's C call sequence, compiled against include/lqr.h and linked to liblqr-hip.so"""
    exe = build("engine", glib)
    for name, args, kw in cases():
        c = compare(exe, engine, tmp_path, name, args, kw)
        ref = H.run_case(oracle, args[0], args[1], args[2], progress=True, **kw)
        assert np.array_equal(c["image"], ref["image"]), name + ": C replay on the engine differs from the oracle"


# ---- the interactive path (render_interactive / render_flatten / render_dump_vmap, src/render.c:465-759) in C --------
STEPS = [("r", 120, 100), ("d",), ("r", 130, 100), ("r", 100, 100), ("d",), ("r", 150, 100), ("r", 140, 90), ("d",), ("f",), ("d",),
         ("r", 120, 80), ("r", 140, 90), ("f",), ("f",), ("r", 139, 90), ("d",), ("r", 170, 90), ("d",)]


def steps_arg():
    return "--steps=" + ",".join("r%dx%d" % (s[1], s[2]) if s[0] == "r" else s[0] for s in STEPS)


def parse_session(path, n_aux):
    r = Reader(path)
    n = r.i()
    recs = []
    for _ in range(n):
        op = chr(r.i())
        rec = {"op": op}
        if op == "d":
            w, h, depth, orientation = r.i(), r.i(), r.i(), r.i()
            rec.update(depth=depth, orientation=orientation, data=r.arr(np.int32, (h, w)))
        else:
            rec["ret"] = r.i()
            rec["state"] = [r.i() for _ in range(7)]        # ref_w ref_h orientation depth enl_step*1000 width height
            rec["nlines"] = r.i()
            w, h, bpp = r.i(), r.i(), r.i()
            rec["image"] = r.arr(np.uint8, (h, w, bpp))
            rec["aux"] = []
            for _ in range(n_aux):
                r.i()
                w, h, bpp = r.i(), r.i(), r.i()
                rec["aux"].append(r.arr(np.uint8, (h, w, bpp)))
        recs.append(rec)
    r.i(); r.i(); r.i()                                     # progress call counts
    assert r.o == len(r.b), "trailing bytes in the session output"
    return recs


def python_session(api, img, kw):
    """the same session through ctypes (tests/harness.py's carver set-up, then call by call)"""
    c, v = H.init_carver(api, img, img.shape[1], img.shape[0], progress=True, **kw)
    recs = []
    for s in STEPS:
        if s[0] == "d":
            vm = c.vmap_dump()
            recs.append(dict(op="d", depth=vm["depth"], orientation=vm["orientation"], data=vm["data"]))
            continue
        ret = c.resize(s[1], s[2]) if s[0] == "r" else c.flatten()
        g = c.getters()
        image, nlines = c.read_scanlines()
        recs.append(dict(op=s[0], ret=ret, nlines=nlines, image=image, aux=[a.read_scanlines()[0] for a in c.aux],
                         state=[g["ref_width"], g["ref_height"], g["orientation"], g["depth"], int(g["enl_step"] * 1000 + 0.5),
                                g["width"], g["height"]]))
    c.destroy()
    return recs


def same_session(a, b, what):
    assert len(a) == len(b), what
    for i, (x, y) in enumerate(zip(a, b)):
        assert x["op"] == y["op"], (what, i)
        if x["op"] == "d":
            assert (x["depth"], x["orientation"]) == (y["depth"], y["orientation"]), (what, i)
            assert np.array_equal(x["data"], y["data"]), "%s: seam map of step %d differs" % (what, i)
        else:
            assert x["ret"] == y["ret"] == L.LQR_OK, (what, i)
            assert list(x["state"]) == list(y["state"]), (what, i, x["state"], y["state"])
            assert x["nlines"] == y["nlines"] and np.array_equal(x["image"], y["image"]), "%s: image of step %d differs" % (what, i)
            assert len(x["aux"]) == len(y["aux"])
            for p, q in zip(x["aux"], y["aux"]):
                assert np.array_equal(p, q), "%s: attached layer of step %d differs" % (what, i)


def session_cases():
    img = D.photo_like(140, 100, 72)
    yield "plain", img, {}
    yield "masks_aux", img, dict(pres=D.ellipse_mask(140, 100), disc=D.band_mask(140, 100, 20, 45), rigmask=D.top_half_mask(140, 100),
                                 rigidity=3.0, resize_aux_layers=True, no_disc_on_enlarge=False)


def run_session(exe, tmp_path, name, img, kw, lib_args=()):
    case, outp = str(tmp_path / (name + ".case")), str(tmp_path / (name + ".sess"))
    write_case(case, img, img.shape[1], img.shape[0], **kw)
    subprocess.check_call([exe] + list(lib_args) + [steps_arg(), case, outp])
    n_aux = sum(1 for k in ("pres", "disc", "rigmask") if kw.get(k) is not None) if kw.get("resize_aux_layers") else 0
    return parse_session(outp, n_aux)


@pytest.mark.parametrize("glib", [False, True], ids=["builtin_typedefs", "glib_typedefs"])
def test_c_interactive_session_against_the_oracle(oracle, tmp_path, glib):
    exe = build("oracle", glib)
    for name, img, kw in session_cases():
        same_session(run_session(exe, tmp_path, name, img, kw), python_session(oracle, img, kw), "oracle, C vs ctypes, " + name)


@pytest.mark.gpu
@pytest.mark.parametrize("glib", [False, True], ids=["builtin_typedefs", "glib_typedefs"])
def test_c_interactive_session_against_the_engine(engine, oracle, tmp_path, glib):
    """resize inside the cached map, beyond it, flatten, lqr_vmap_dump + lqr_vmap_get_* on a caller-owned map, the getters
    after every step, read-out after every step -- compiled as C against include/lqr.h, linked to liblqr-hip.so"""
    exe = build("engine", glib)
    for name, img, kw in session_cases():
        c = run_session(exe, tmp_path, name, img, kw)
        same_session(c, python_session(engine, img, kw), "engine, C vs ctypes, " + name)
        same_session(c, python_session(oracle, img, kw), "engine (C) vs oracle, " + name)


# ---- one binary, any library: every lqr_* entry point resolved with dlsym -------------------------------------------
def test_c_replay_dlopen_the_oracle(oracle, tmp_path):
    """-DREPLAY_DLOPEN build: links to no carving library, resolves all 48 declared functions from --lib (the oracle's
    exports carry the prefix "o"); tests/test_real_liblqr.py points the same binary at a genuine liblqr-1"""
    exe = build("dlopen", False)
    lib_args = ["--lib=" + ORACLE_SO, "--prefix=o"]
    for name, args, kw in cases():
        compare(exe, oracle, tmp_path, name, args, kw, lib_args)
    for name, img, kw in session_cases():
        same_session(run_session(exe, tmp_path, name, img, kw, lib_args), python_session(oracle, img, kw), "dlopen oracle, " + name)


@pytest.mark.gpu
def test_c_replay_dlopen_the_engine(engine, tmp_path):
    exe = build("dlopen", True)
    lib_args = ["--lib=" + ENGINE_SO]
    for name, args, kw in cases():
        compare(exe, engine, tmp_path, name, args, kw, lib_args)
    for name, img, kw in session_cases():
        same_session(run_session(exe, tmp_path, name, img, kw, lib_args), python_session(engine, img, kw), "dlopen engine, " + name)
