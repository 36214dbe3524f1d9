"""Round 5 (-m gpu).
* A lock-step group of 32 or more delta_x = 2 / rigidity-mask carvers that fits the tiled kernels' residency bound is NOT split
  over sub-batch streams (shared batches never run the persistent kernels: ADVICE r4 -- such a group fell to the
  one-wave-per-image kernels, ~30x slower, now that the library sets GPU_MAX_HW_QUEUES=8 itself); since k_band_levels has
  general instantiations such a group runs on them (or on the full-width tiled kernels), never on k_band_update.
* k_band_levels with delta_x 2 .. 4 and rigidity masks: lock-step batches of 9 images per variant against the oracle, and whole
  resizes in both directions with the kernel forced (update mode 5) on single images.
* k_band_levels' two copies of every hand-over word: with the near (L2-resident) copy off, and with an image's slots placed on
  different XCDs (where the near copy is never seen and every fourth poll, of the write-through copy, carries the protocol).
* k_dp_tile_p's tile numbering (the workgroups of one XCD hold consecutive tiles) and near copies: round 4's numbering and "no near
  copies" (lqrhip_dp_tile_debug 2 / 1) give the same results on a single image and on a group of 4.
* lqrhip_moved_bytes: the bytes the carves had to move, as k_vpath* counts them, against a count made from the seam maps.
"""
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


@pytest.mark.parametrize("variant", ["delta2", "rigmask"])
def test_general_group_of_36_small_images_stays_off_the_one_wave_kernels(oracle, engine, variant):
    lib = engine.lib
    lib.lqrhip_sub_batches.argtypes = [ctypes.c_int]
    lib.lqrhip_set_sub_batches.argtypes = [ctypes.c_int]
    lib.lqrhip_general_batch_limit.argtypes = [ctypes.c_int]
    w, h, n = 150, 70, 36
    kw = dict(delta2=dict(delta_x=2), rigmask=dict(rigidity=5.0))[variant]
    rigm = D.top_half_mask(w, h) if variant == "rigmask" else None
    assert lib.lqrhip_general_batch_limit(w) >= n, "the group must fit the residency bound for this test to mean anything"
    imgs = [D.photo_like(w, h, 5100 + i) if i % 3 else D.noise(w, h, 5100 + i) for i in range(n)]
    lib.lqrhip_set_sub_batches(4)            # what a plain group of this size gets (and what the automatic choice is with 8 queues)
    try:
        assert lib.lqrhip_sub_batches(n) == 4
        cs = []
        for im in imgs:
            c = L.Carver(engine, im, delta_x=kw.get("delta_x", 1), rigidity=(3 * kw.get("rigidity", 0.0) if rigm is not None else kw.get("rigidity", 0.0)))
            if rigm is not None:
                assert c.rigmask_add(rigm) == L.LQR_OK
            cs.append(c.configure())
        lib.lqrhip_prof_reset(); lib.lqrhip_prof_enable(1)
        assert L.resize_batch(engine, cs, w - 21, h - 6) == L.LQR_OK
        lib.lqrhip_prof_enable(0)
        assert prof_launches(lib, "band_levels") + prof_launches(lib, "dp_update_tiled") > 0 and prof_launches(lib, "band_update") == 0, \
            "the group fell to the one-wave band kernel"
        for c, im in zip(cs, imgs):
            ref = H.run_case(oracle, im, w - 21, h - 6, rigmask=rigm, **kw)
            assert np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"])
            assert np.array_equal(c.read_image(), ref["image"])
        for c in cs:
            c.destroy()
    finally:
        lib.lqrhip_prof_enable(0)
        lib.lqrhip_set_sub_batches(0)


def test_moved_bytes_equal_the_shorter_side_of_every_seam(oracle, engine):
    """One 400x90 image, 30 vertical seams, no side switch (one full DP at the start, 30 incremental updates): the engine's count
    must be 18 B x (pixels on the side of each seam that is shorter over the whole image), as read off the seam map."""
    lib = engine.lib
    lib.lqrhip_moved_bytes.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    w, h, ns = 400, 90, 30
    img = D.photo_like(w, h, 77)
    c = L.Carver(engine, img).configure(switch_freq=0)
    z = ctypes.c_ulonglong(0)
    assert lib.lqrhip_moved_bytes(ctypes.byref(z), 1) == 0
    assert c.resize(w - ns, h) == L.LQR_OK
    got = ctypes.c_ulonglong(0)
    assert lib.lqrhip_moved_bytes(ctypes.byref(got), 1) == 0
    vm = c.vmap_dump()["data"].astype(np.int64)          # 0 = never carved, else the seam's level (in carving order, from the deepest)
    c.destroy()
    levels = sorted(set(int(v) for v in np.unique(vm)) - {0})
    assert len(levels) == ns
    # replay the carving order: the seam carved k-th has the k-th value in the order the engine numbered them; its x in the frame of
    # its own time = the number of not-yet-carved pixels to its left
    order = sorted(levels, reverse=True) if vm.max() > 0 else levels
    # which end of the value range is carved first does not matter for the SUM below only if we replay in the true order: try both
    def total(order):
        alive = np.ones((h, w), bool)
        tot = 0
        cur_w = w
        for lv in order:
            xs = []
            for y in range(h):
                col = int(np.flatnonzero(vm[y] == lv)[0])
                xs.append(int(alive[y, :col].sum()))
                alive[y, col] = False
            left = sum(xs); right = h * (cur_w - 1) - left
            tot += 18 * (left if 2 * left < h * (cur_w - 1) else right)
            cur_w -= 1
        return tot
    assert got.value in (total(order), total(order[::-1])), (got.value, total(order), total(order[::-1]))


GENERAL = {"delta2": dict(delta_x=2), "delta3": dict(delta_x=3), "delta4-rigidity": dict(delta_x=4, rigidity=3.0),
           "rigmask": dict(rigidity=5.0, rigmask=True), "delta2-rigmask": dict(delta_x=2, rigidity=2.0, rigmask=True)}


@pytest.mark.parametrize("variant", sorted(GENERAL))
def test_band_levels_general_batch_of_9(oracle, engine, variant):
    """a group of 9 (>= 8: the engine's own choice is k_band_levels) with delta_x 2 .. 4 / a rigidity mask, every image against the oracle"""
    lib = engine.lib
    w, h, n = 700, 260, 9
    v = dict(GENERAL[variant])
    rigm = D.top_half_mask(w, h) if v.pop("rigmask", False) else None
    imgs = [D.photo_like(w, h, 7100 + i) if i % 2 else D.noise(w, h, 7100 + i) for i in range(n)]
    cs = []
    for im in imgs:
        c = L.Carver(engine, im, delta_x=v.get("delta_x", 1), rigidity=(3 * v.get("rigidity", 0.0) if rigm is not None else v.get("rigidity", 0.0)))
        if rigm is not None:
            assert c.rigmask_add(rigm) == L.LQR_OK
        cs.append(c.configure())
    lib.lqrhip_prof_reset(); lib.lqrhip_prof_enable(1)
    try:
        assert L.resize_batch(engine, cs, w - 50, h - 20) == L.LQR_OK
    finally:
        lib.lqrhip_prof_enable(0)
    assert prof_launches(lib, "band_levels") > 0 and prof_launches(lib, "band_update") == 0
    for c, im in zip(cs, imgs):
        ref = H.run_case(oracle, im, w - 50, h - 20, rigmask=rigm, **v)
        assert np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"])
        assert np.array_equal(c.read_image(), ref["image"])
    for c in cs:
        c.destroy()


@pytest.mark.parametrize("variant", sorted(GENERAL))
@pytest.mark.parametrize("slots", [2, 5, 16])
def test_band_levels_general_forced_on_single_images(oracle, engine, variant, slots):
    """update mode 5 with few and many slots (few: two tiles per slot and level, three = the image stops and the sweep takes over)"""
    lib = engine.lib
    lib.lqrhip_set_update_mode.argtypes = [ctypes.c_int]
    lib.lqrhip_set_band_levels.argtypes = [ctypes.c_int]
    w, h = 900, 330
    v = dict(GENERAL[variant])
    rigm = D.ellipse_mask(w, h) if v.pop("rigmask", False) else None
    img = D.photo_like(w, h, 88)
    kw = dict(v, pres=D.ellipse_mask(w, h), output_seams=True)
    lib.lqrhip_set_update_mode(5); lib.lqrhip_set_band_levels(slots)
    try:
        a = H.run_case(oracle, img, w - 70, h - 25, rigmask=rigm, **kw)
        b = H.run_case(engine, img, w - 70, h - 25, rigmask=rigm, **kw)
        H.assert_same(a, b, "levels general %s, %d slots" % (variant, slots))
    finally:
        lib.lqrhip_set_update_mode(-1); lib.lqrhip_set_band_levels(-1)


@pytest.mark.parametrize("dbg", [1, 2])
def test_dp_tile_p_tile_numbering_and_near_copies(oracle, engine, dbg):
    """1: no near copies; 2: tile = workgroup index (neighbours on different XCDs, no near copies) -- a single image in both directions
    (the plug-in's own call shape: k_dp_tile_p<UPDATE> per seam, E5 at the side switches) and a lock-step group of 4"""
    lib = engine.lib
    lib.lqrhip_dp_tile_debug.argtypes = [ctypes.c_int]
    lib.lqrhip_dp_tile_debug(dbg)
    try:
        img = D.photo_like(1500, 400, 77)
        a = H.run_case(oracle, img, 1440, 380, output_seams=True)
        b = H.run_case(engine, img, 1440, 380, output_seams=True)
        H.assert_same(a, b, "k_dp_tile_p, debug %d" % dbg)
        w, h, n = 1000, 260, 4
        imgs = [D.photo_like(w, h, 9300 + i) if i % 2 else D.noise(w, h, 9300 + i) for i in range(n)]
        cs = [L.Carver(engine, im).configure() for im in imgs]
        lib.lqrhip_prof_reset(); lib.lqrhip_prof_enable(1)
        try:
            assert L.resize_batch(engine, cs, w - 50, h) == L.LQR_OK
        finally:
            lib.lqrhip_prof_enable(0)
        assert prof_launches(lib, "dp_update_tiled") > 0
        for c, im in zip(cs, imgs):
            ref = H.run_case(oracle, im, w - 50, h)
            assert np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"])
            assert np.array_equal(c.read_image(), ref["image"])
        for c in cs:
            c.destroy()
    finally:
        lib.lqrhip_dp_tile_debug(0)


@pytest.mark.parametrize("dbg", [4, 8, 12])
def test_band_levels_handover_copies_and_slot_placement(oracle, engine, dbg):
    """4: no near copy; 8: the slots of an image on different XCDs (near copies written, never seen); 12: both -- a lock-step
    group of 9 and a forced single image against the oracle.  Nothing but speed may depend on where the slots sit."""
    lib = engine.lib
    lib.lqrhip_set_update_mode.argtypes = [ctypes.c_int]
    lib.lqrhip_set_band_levels.argtypes = [ctypes.c_int]
    lib.lqrhip_band_levels_debug.argtypes = [ctypes.c_int]
    w, h, n = 1100, 300, 9
    imgs = [D.photo_like(w, h, 9100 + i) if i % 2 else D.noise(w, h, 9100 + i) for i in range(n)]
    lib.lqrhip_band_levels_debug(dbg)
    try:
        cs = [L.Carver(engine, im).configure() for im in imgs]
        lib.lqrhip_prof_reset(); lib.lqrhip_prof_enable(1)
        try:
            assert L.resize_batch(engine, cs, w - 60, h - 20) == L.LQR_OK
        finally:
            lib.lqrhip_prof_enable(0)
        assert prof_launches(lib, "band_levels") > 0
        for c, im in zip(cs, imgs):
            ref = H.run_case(oracle, im, w - 60, h - 20)
            assert np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"])
            assert np.array_equal(c.read_image(), ref["image"])
        for c in cs:
            c.destroy()
        lib.lqrhip_set_update_mode(5); lib.lqrhip_set_band_levels(5)
        img = D.photo_like(1400, 420, 31)
        a = H.run_case(oracle, img, 1330, 400, output_seams=True)
        b = H.run_case(engine, img, 1330, 400, output_seams=True)
        H.assert_same(a, b, "levels, debug %d" % dbg)
    finally:
        lib.lqrhip_band_levels_debug(0); lib.lqrhip_set_update_mode(-1); lib.lqrhip_set_band_levels(-1)


@pytest.mark.parametrize("w,h", [(300, 160), (1400, 700), (40, 9), (2500, 31), (700, 57), (3000, 29), (64, 1), (900, 2), (3840, 300)])
def test_backtrack_chunk_boundaries(oracle, engine, w, h):
    """delta_x = 1 (k_vpath1<1>, 28-row chunks): heights around the chunk size (29 = one full chunk, 31, 57 = two, 9 = a partial one,
    1 and 2 = none / one row) and rows wider than its 256-column window, against the oracle"""
    img = D.photo_like(w, h, w + 3 * h)
    nw = w - min(30, w // 3)
    H.assert_same(H.run_case(oracle, img, nw, h), H.run_case(engine, img, nw, h), "backtrack %dx%d" % (w, h))
