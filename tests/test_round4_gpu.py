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


def test_band_tiles_seeded_cases_with_forced_tile_counts():
    """scripts/fuzz_tiles.py: the multi-CU band update with 1..12 tiles per image -- sets narrower than the band (coverage
    test, abort at the edge of the set and hand-over to the full-width sweep), sets wider than the image, both tie rules,
    rigidity; seam maps, pixels and the DP planes after the last incremental update against the oracle"""
    import fuzz_common as FC
    FC.run_script("fuzz_tiles.py", [0, 4400], {}, 400)          # count-bounded: all 400 cases must have run


def test_band_tiles_batch_of_12_images(oracle, engine):
    """a lock-step batch of the size the engine itself sends to k_band_tiles (8 to ~40 images): every image against the oracle"""
    import ctypes
    lib = engine.lib
    n, w, h = 12, 1200, 330
    imgs = [D.photo_like(w, h, 1200 + i) if i % 2 else D.noise(w, h, 1200 + i) for i in range(n)]
    cs = [L.Carver(engine, im).configure() for im in imgs]
    lib.lqrhip_prof_reset(); lib.lqrhip_prof_enable(1)
    assert L.resize_batch(engine, cs, w - 60, h) == L.LQR_OK
    lib.lqrhip_prof_enable(0)
    st = (ctypes.c_ulonglong * 8)()
    for c, im in zip(cs, imgs):
        ref = H.run_case(oracle, im, w - 60, h)
        assert np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"])
        assert np.array_equal(c.read_image(), ref["image"])
    for c in cs:
        c.destroy()


@pytest.mark.parametrize("seed", [607398534, 7001])        # the first one is the image of the fuzz case below
def test_band_tiles_chains_of_reserve_tiles(oracle, engine, seed):
    """3 base tiles and 9 reserves on a tall noise image with a discard band: the band of changes walks outward through
    several reserve tiles, one waking the next.  A reserve tile may only give up waiting when every tile that ever started
    has ended (k_band_tiles's header count): the first version left when the BASE tiles were done, and a request made two
    hops out in the second-to-last block then waited for a tile that had gone (fuzz seed 30311 case 1308: time-out)."""
    lib = engine.lib
    lib.lqrhip_set_update_mode.argtypes = [ctypes.c_int]; lib.lqrhip_set_band_tiles.argtypes = [ctypes.c_int]; lib.lqrhip_set_band_tiles_reserve.argtypes = [ctypes.c_int]
    w, h = 1125, 834
    img = D.noise(w, h, seed, channels=2)
    kw = dict(nrg_func=3, switch_freq=9, res_order=1, pres=D.ellipse_mask(w, h), disc=D.band_mask(w, h, w // 5, w // 3))
    lib.lqrhip_set_update_mode(4); lib.lqrhip_set_band_tiles(12); lib.lqrhip_set_band_tiles_reserve(9)
    try:
        ca, _ = H.init_carver(oracle, img, w - 46, h, **kw)
        assert ca.resize(w - 46, h) == L.LQR_OK
        vref, iref = ca.vmap_dump()["data"], ca.read_image()
        for rep in range(10):            # (who leaves first is a matter of timing: the old exit failed one run in a few)
            cb, _ = H.init_carver(engine, img, w - 46, h, **kw)
            assert cb.resize(w - 46, h) == L.LQR_OK, rep
            assert np.array_equal(vref, cb.vmap_dump()["data"])
            assert np.array_equal(iref, cb.read_image())
            if rep < 9:
                cb.destroy()
        ca.destroy(); cb.destroy()
    finally:
        lib.lqrhip_set_update_mode(-1); lib.lqrhip_set_band_tiles(-1); lib.lqrhip_set_band_tiles_reserve(-1)
