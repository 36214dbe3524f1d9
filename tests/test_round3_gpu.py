"""Round 3 (-m gpu): delta_x = 2 and rigidity masks on the tiled full-width kernels (k_dp_tile_p's general
instantiations, 5-neighbour rows / per-pixel rigidity factor), which round 2 sent to the one-wave-per-image band kernel
and the one-workgroup-per-image sweep; plus the full-size configs against the oracle.
"""
import ctypes

import numpy as np
import pytest

import datasets as D
import harness as H
import lqr_ctypes as L

pytestmark = pytest.mark.gpu

VARIANTS = {
    "rigmask": dict(delta_x=1, rigidity=7.0, rigmask=True),
    "delta2": dict(delta_x=2, rigidity=0.0, rigmask=False),
    "delta2-rigidity": dict(delta_x=2, rigidity=5.0, rigmask=False),
    "delta2-rigmask": dict(delta_x=2, rigidity=9.0, rigmask=True),
    # round 4: delta_x 3 and 4 on the same kernels (8-row blocks, the parents reach two lanes out)
    "delta3": dict(delta_x=3, rigidity=0.0, rigmask=False),
    "delta4-rigidity": dict(delta_x=4, rigidity=5.0, rigmask=False),
    "delta3-rigmask": dict(delta_x=3, rigidity=9.0, rigmask=True),
    "delta4": dict(delta_x=4, rigidity=0.0, rigmask=False),
}


def _kw(v, w, h, switch_freq):
    kw = dict(delta_x=v["delta_x"], rigidity=v["rigidity"], switch_freq=switch_freq)
    if v["rigmask"]:
        # a mask with structure in both directions: a ramp times the top-half step (values 0..255, alpha 255)
        m = np.zeros((h, w, 4), np.uint8)
        m[:, :, :3] = (np.arange(w, dtype=np.int64)[None, :, None] * 255 // max(w - 1, 1)).astype(np.uint8)
        m[h // 2:, :, :3] //= 3
        m[:, :, 3] = 255
        kw["rigmask"] = m
    return kw


@pytest.mark.parametrize("variant", sorted(VARIANTS))
@pytest.mark.parametrize("kind,w,h,n_seams", [("photo", 700, 300, 40), ("noise", 1500, 150, 30), ("flat", 400, 220, 35), ("noise", 131, 517, 25)])
def test_dp_planes_general_tiled(oracle, engine, variant, kind, w, h, n_seams):
    """en, m and the back pointers after the last incremental update, bit for bit, with the incremental updates on the
    tiled kernel (auto), on the generic band kernel + sweep (band) and on k_band_levels (levels): several tiles wide (64 own columns each), more
    rows than one block (32 / delta_x rows), tie-heavy input included; switch_freq 0 keeps every update incremental"""
    v = VARIANTS[variant]
    img = {"photo": D.photo_like, "noise": D.noise, "flat": D.flat_blocks}[kind](w, h, w + h)
    kw = _kw(v, w, h, 0)
    engine.lib.lqrhip_set_update_mode.argtypes = [ctypes.c_int]
    oracle.lqrx_set_debug(1)
    c, _ = H.init_carver(oracle, img, w - n_seams, h, **kw)
    assert c.resize(w - n_seams, h) == L.LQR_OK
    ea, ma, da = c.debug_snapshot()
    oracle.lqrx_set_debug(0)
    c.destroy()
    try:
        for mode in (-1, 0, 5):          # 5: k_band_levels' general instantiations (round 5)
            engine.lib.lqrhip_set_update_mode(mode)
            engine.lqrx_set_debug(1)
            c, _ = H.init_carver(engine, img, w - n_seams, h, **kw)
            assert c.resize(w - n_seams, h) == L.LQR_OK
            eb, mb, db = c.debug_snapshot()
            engine.lqrx_set_debug(0)
            c.destroy()
            assert np.array_equal(ea, eb) and np.array_equal(ma, mb) and np.array_equal(da[1:], db[1:]), (variant, kind, w, h, mode)
    finally:
        engine.lqrx_set_debug(0)
        engine.lib.lqrhip_set_update_mode(-1)


@pytest.mark.parametrize("variant", sorted(VARIANTS))
@pytest.mark.parametrize("switch_freq", [2, 1000])
def test_general_tiled_both_tie_rules_and_directions(oracle, engine, variant, switch_freq):
    """whole resizes through the C ABI, both directions (the rigidity table is rescaled on transpose), the side switching
    after every seam (switch_freq 1000: a full DP per seam, both tie rules) or twice per rescale; masks, seam maps"""
    v = VARIANTS[variant]
    w, h = 420, 260
    img = D.photo_like(w, h, 77)
    kw = _kw(v, w, h, switch_freq)
    kw.update(pres=D.ellipse_mask(w, h), disc=D.band_mask(w, h, 60, 110), output_seams=True)
    a = H.run_case(oracle, img, w - 45, h - 30, **kw)
    b = H.run_case(engine, img, w - 45, h - 30, **kw)
    H.assert_same(a, b, "general tiled %s freq %d" % (variant, switch_freq))


def test_general_tiled_batch(oracle, engine):
    """a lock-step batch (3 images) with delta_x 2 and a rigidity mask: one persistent grid covers every image's tiles"""
    w, h = 500, 180
    imgs = [D.noise(w, h, 31 + i) for i in range(3)]
    kw = _kw(VARIANTS["delta2-rigmask"], w, h, 2)
    cs = [H.init_carver(engine, im, w - 40, h, **kw)[0] for im in imgs]
    assert L.resize_batch(engine, cs, w - 40, h) == L.LQR_OK
    for im, c in zip(imgs, cs):
        ref = H.run_case(oracle, im, w - 40, h, **kw)
        assert np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"])
        assert np.array_equal(c.read_image(), ref["image"])
        c.destroy()


def test_sub_batch_streams_are_the_default_for_large_groups(oracle, engine):
    """lqrhip_sub_batches: 4 streams for 49 carvers and more, 2 for 32 to 48 (round 5) when the process says it has the hardware queues
    (GPU_MAX_HW_QUEUES >= 8), else one; then a 34-image group through that default path, image by image
    against the oracle.  (The variable only steers the engine's choice here: HIP read it, or its absence, long ago.  It is
    changed in the C environment -- where the library set it when it was loaded, tests/test_queue_env.py -- not in
    os.environ, which does not see what C code sets.)"""
    lib = engine.lib
    lib.lqrhip_set_sub_batches.argtypes = [ctypes.c_int]
    lib.lqrhip_sub_batches.argtypes = [ctypes.c_int]
    libc = ctypes.CDLL(None)
    libc.getenv.restype = ctypes.c_char_p
    before = libc.getenv(b"GPU_MAX_HW_QUEUES")
    lib.lqrhip_set_sub_batches(0)
    try:
        libc.unsetenv(b"GPU_MAX_HW_QUEUES")
        assert [lib.lqrhip_sub_batches(n) for n in (1, 31, 32, 64)] == [1, 1, 1, 1]
        libc.setenv(b"GPU_MAX_HW_QUEUES", b"4", 1)
        assert lib.lqrhip_sub_batches(64) == 1
        libc.setenv(b"GPU_MAX_HW_QUEUES", b"8", 1)
        assert [lib.lqrhip_sub_batches(n) for n in (1, 31, 32, 48, 49, 64)] == [1, 1, 2, 2, 4, 4]
        # a forced count, then the 34-image group on the engine's own choice (2 streams, k_band_levels)
        lib.lqrhip_set_sub_batches(2)
        assert [lib.lqrhip_sub_batches(n) for n in (3, 4, 64)] == [1, 2, 2]
        lib.lqrhip_set_sub_batches(0)
        w, h = 260, 90
        imgs = [D.noise(w, h, 400 + i) if i % 3 else D.photo_like(w, h, 400 + i) for i in range(34)]
        cs = [L.Carver(engine, im).configure() for im in imgs]
        assert L.resize_batch(engine, cs, w - 25, h - 10) == L.LQR_OK
        for i, (im, c) in enumerate(zip(imgs, cs)):
            if i % 4 == 0 or i == 33:
                ref = H.run_case(oracle, im, w - 25, h - 10)
                assert np.array_equal(c.read_image(), ref["image"]), i
                assert np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"]), i
            c.destroy()
    finally:
        if before is None:
            libc.unsetenv(b"GPU_MAX_HW_QUEUES")
        else:
            libc.setenv(b"GPU_MAX_HW_QUEUES", before, 1)


# ---- BASELINE.json's large configs against the oracle at FULL size (round 2 checked them through properties only) ----
def test_config3_full_size_vs_oracle(oracle, engine):
    """config 3: 3840x2160 -> 3340x1660 (500 vertical + 500 horizontal seams): the engine's image and both seam maps
    equal the oracle's (about 20 s of CPU)"""
    img = D.noise(3840, 2160, 3)
    a = H.run_case(oracle, img, 3340, 1660, output_seams=True)
    b = H.run_case(engine, img, 3340, 1660, output_seams=True)
    H.assert_same(a, b, "config 3 at full size")


def test_config5_full_size_vs_oracle(oracle, engine):
    """config 5: 7680x4320, preservation ellipse (+1000), discard band (-1000), rigidity 10, delta_x 1, 1000 seams:
    image and seam map equal the oracle's (about 50 s of CPU)"""
    w, h, n = 7680, 4320, 1000
    img = D.noise(w, h, 5)
    kw = dict(pres=D.ellipse_mask(w, h), disc=D.band_mask(w, h, 1500, 2100), rigidity=10.0)
    a = H.run_case(oracle, img, w - n, h, **kw)
    b = H.run_case(engine, img, w - n, h, **kw)
    H.assert_same(a, b, "config 5 at full size")


@pytest.mark.parametrize("variant", ["delta2", "rigmask"])
def test_config5_variants_reduced_vs_oracle(oracle, engine, variant):
    """config 5's variants (delta_x 2; rigidity mask => rigidity x 3, render.c:784-787) at 1/4 linear scale, 250 seams"""
    w, h, n = 1920, 1080, 250
    img = D.noise(w, h, 5)
    kw = dict(pres=D.ellipse_mask(w, h), disc=D.band_mask(w, h, 375, 525), rigidity=10.0)
    if variant == "delta2":
        kw["delta_x"] = 2
    else:
        kw["rigmask"] = D.top_half_mask(w, h)
    a = H.run_case(oracle, img, w - n, h, **kw)
    b = H.run_case(engine, img, w - n, h, **kw)
    H.assert_same(a, b, "config 5 variant " + variant)
