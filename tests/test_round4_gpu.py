"""Round 4 (-m gpu): large lock-step batches of delta_x = 2 / rigidity-mask carvers are carved group after group on the
tiled kernels instead of falling to the one-wave-per-image kernels (host/lqr_carver.c lqrx_carver_resize_batch,
lqrhip_general_batch_limit); host transfers through the ring of pinned buffers (odd sizes, both directions)."""
import ctypes

import numpy as np
import pytest

import datasets as D
import harness as H
import lqr_ctypes as L

pytestmark = pytest.mark.gpu


def prof_launches(lib, name):
    ms, n, by = ctypes.c_double(0), ctypes.c_longlong(0), ctypes.c_double(0)
    lib.lqrhip_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(n), ctypes.byref(by))
    return n.value


@pytest.mark.parametrize("variant", ["delta2", "rigmask", "delta2-rigidity", "delta4"])
def test_large_general_batches_run_group_after_group_on_the_tiled_kernels(oracle, engine, variant):
    lib = engine.lib
    lib.lqrhip_set_dp_persistent_limit.argtypes = [ctypes.c_int]
    lib.lqrhip_general_batch_limit.argtypes = [ctypes.c_int]
    w, h, n = 520, 140, 7
    kw = dict(delta2=dict(delta_x=2), rigmask=dict(rigidity=6.0), delta4=dict(delta_x=4), **{"delta2-rigidity": dict(delta_x=2, rigidity=4.0)})[variant]
    rigm = D.top_half_mask(w, h) if variant == "rigmask" else None
    imgs = [D.photo_like(w, h, 900 + i) for i in range(n)]
    tiles = (w + 63) // 64
    lib.lqrhip_set_dp_persistent_limit(3 * tiles)            # three images' worth of tiles: groups of 3, 3, 1
    try:
        assert lib.lqrhip_general_batch_limit(w) == 3
        cs = []
        for im in imgs:
            c = L.Carver(engine, im, delta_x=kw.get("delta_x", 1), rigidity=(3 * kw.get("rigidity", 0.0) if rigm is not None else kw.get("rigidity", 0.0)))
            if rigm is not None:
                assert c.rigmask_add(rigm) == L.LQR_OK
            cs.append(c.configure())
        lib.lqrhip_prof_reset(); lib.lqrhip_prof_enable(1)
        assert L.resize_batch(engine, cs, w - 37, h - 11) == L.LQR_OK
        lib.lqrhip_prof_enable(0)
        assert prof_launches(lib, "dp_update_tiled") > 0 and prof_launches(lib, "band_update") == 0, "the batch fell to the band kernels"
        for c, im in zip(cs, imgs):
            ref = H.run_case(oracle, im, w - 37, h - 11, rigmask=rigm, **kw)
            assert np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"])
            assert np.array_equal(c.read_image(), ref["image"])
        for c in cs:
            c.destroy()
    finally:
        lib.lqrhip_prof_enable(0)
        lib.lqrhip_set_dp_persistent_limit(-1)


@pytest.mark.parametrize("w,h,ch", [(1, 1, 1), (7, 5, 3), (1031, 517, 4), (2049, 1023, 2), (4099, 1037, 4)])
def test_host_transfers_odd_sizes(engine, w, h, ch):
    """upload (lqr_carver_new) and read-out (scan lines and the whole-image read) of sizes that are not multiples of the
    4 MiB staging chunks, below one chunk, and several chunks long; twice, so that the ring's slots are reused"""
    rng = np.random.default_rng(w * 7 + h)
    for _ in range(2):
        img = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
        c = L.Carver(engine, img)
        assert np.array_equal(c.read_image(), img)
        lines, n = c.read_scanlines()
        assert n == h and np.array_equal(lines, img)
        assert np.array_equal(c.read_image(), img)
        c.destroy()


def test_batch_of_12_images_on_the_engines_own_choice(oracle, engine):
    """a lock-step batch of the size the engine sends to the multi-CU band update (8 and more images): every image against the oracle"""
    n, w, h = 12, 1200, 330
    imgs = [D.photo_like(w, h, 1200 + i) if i % 2 else D.noise(w, h, 1200 + i) for i in range(n)]
    cs = [L.Carver(engine, im).configure() for im in imgs]
    assert L.resize_batch(engine, cs, w - 60, h) == L.LQR_OK
    for c, im in zip(cs, imgs):
        ref = H.run_case(oracle, im, w - 60, h)
        assert np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"])
        assert np.array_equal(c.read_image(), ref["image"])
    for c in cs:
        c.destroy()
