"""Round 6 (-m gpu): the parallel backtrack (k_vp_maps / k_vp_solve, csrc/k_backtrack.hip) -- the engine's choice for one to three images of
1000 rows and more -- against the oracle and against the one-wave walk k_vpath1 it replaces there: shapes around the chunk size
(56 / delta_x rows), the 256-column tiles of the map kernel and the stage length of the solver (20 chunks); delta_x 1 .. 4; both
directions; groups; the moved-bytes accounting."""
import ctypes

import numpy as np
import pytest

import datasets as D
import harness as H
import lqr_ctypes as L

pytestmark = pytest.mark.gpu


@pytest.fixture()
def lib(engine):
    lb = engine.lib
    lb.lqrhip_set_vpath_mode.argtypes = [ctypes.c_int, ctypes.c_int]
    yield lb
    lb.lqrhip_set_vpath_mode(-1, 16)


SHAPES = [(40, 2), (40, 3), (300, 56), (300, 57), (300, 58), (255, 113), (256, 114), (257, 169), (600, 449), (600, 450), (1100, 505), (31, 700), (3, 90), (120, 1121), (90, 1122), (200, 2300)]


@pytest.mark.parametrize("w,h", SHAPES)
@pytest.mark.parametrize("delta", [1, 2, 3, 4, 7, 10])
def test_parallel_backtrack_chunk_tile_and_stage_boundaries(oracle, engine, lib, w, h, delta):
    img = D.photo_like(w, h, 600 + w + h) if (w + h) % 2 else D.noise(w, h, 600 + w + h)
    nw, nh = max(2, w - min(12, w // 3)), max(2, h - min(9, h // 4))
    kw = dict(delta_x=delta, rigidity=(2.0 if delta == 3 else 0.0), output_seams=True)
    ref = H.run_case(oracle, img, nw, nh, **kw)
    for mode in (1, 0):
        lib.lqrhip_set_vpath_mode(mode, 0)
        H.assert_same(ref, H.run_case(engine, img, nw, nh, **kw), "%dx%d -> %dx%d delta %d, backtrack mode %d" % (w, h, nw, nh, delta, mode))


@pytest.mark.parametrize("n", [1, 3, 9])
def test_parallel_backtrack_groups_and_moved_bytes(oracle, engine, lib, n):
    """groups of 1, 3 (full-width tiled update) and 9 (k_band_levels) with the parallel backtrack forced; the bytes the carves had to
    move equal what the one-wave walk counts"""
    lib.lqrhip_moved_bytes.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    w, h = 700, 260
    imgs = [D.photo_like(w, h, 800 + i) for i in range(n)]
    moved = {}
    for mode in (1, 0):
        lib.lqrhip_set_vpath_mode(mode, 0)
        cs = [L.Carver(engine, im).configure() for im in imgs]
        z = ctypes.c_ulonglong(0)
        lib.lqrhip_moved_bytes(ctypes.byref(z), 1)
        assert L.resize_batch(engine, cs, w - 45, h - 20) == L.LQR_OK
        lib.lqrhip_moved_bytes(ctypes.byref(z), 1)
        moved[mode] = z.value
        for c, im in zip(cs, imgs):
            ref = H.run_case(oracle, im, w - 45, h - 20)
            assert np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"]) and np.array_equal(c.read_image(), ref["image"]), mode
        for c in cs:
            c.destroy()
    assert moved[0] == moved[1] > 0, moved


def test_parallel_backtrack_config3_size_both_directions(oracle, engine, lib):
    """a 4K image, 60 + 40 seams (39 chunks of 56 rows, 15 column tiles, 2 stages; then 69 chunks, 4 stages): the plug-in's own call shape"""
    img = D.photo_like(3840, 2160, 3)
    ref = H.run_case(oracle, img, 3780, 2120)
    lib.lqrhip_set_vpath_mode(1, 0)
    H.assert_same(ref, H.run_case(engine, img, 3780, 2120), "4K both directions, parallel backtrack")


# ---------------------------------------------------------------- delta_x 5 .. 10 on the tiled kernels
def prof_launches(lb, name):
    ms, n, by = ctypes.c_double(0), ctypes.c_longlong(0), ctypes.c_double(0)
    lb.lqrhip_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(n), ctypes.byref(by))
    return n.value


WIDE_VARIANTS = {
    "plain": {},
    "rigidity": dict(rigidity=5.0),
    "rigmask": dict(rigidity=3.0, rigmask=True),
    "masks": dict(pres=True, disc=True),
}


@pytest.mark.parametrize("delta", [5, 6, 7, 8, 9, 10])
@pytest.mark.parametrize("variant", sorted(WIDE_VARIANTS))
@pytest.mark.parametrize("path", ["auto", "levels", "generic"])
def test_wide_delta_dp_planes_and_seams(oracle, engine, lib, delta, variant, path):
    """delta_x 5 .. 10 -- the upper half of the plug-in's dialog range (src/interface.c:47) -- ran on the one-wave kernels until round 6
    (~30x slower).  Now: k_dp_tile_p's / k_band_levels' general instantiations with blocks of 32 / delta_x rows, the reach of a row
    fetched from up to five lanes away.  After 30 incremental updates the DP planes (energies, cumulative minima, back pointers) are
    bit-identical to the oracle's; whole resizes in both directions; the generic kernels (update mode 3) as the third opinion."""
    lb = lib
    lb.lqrhip_set_update_mode.argtypes = [ctypes.c_int]; lb.lqrhip_set_band_levels.argtypes = [ctypes.c_int]
    w, h = 460, 190
    img = D.photo_like(w, h, 40 + delta) if delta % 2 else D.noise(w, h, 40 + delta)
    v = dict(WIDE_VARIANTS[variant])
    kw = dict(delta_x=delta, switch_freq=0)
    if v.pop("rigmask", False):
        kw["rigmask"] = D.top_half_mask(w, h)
    if v.pop("pres", False):
        kw["pres"] = D.ellipse_mask(w, h)
    if v.pop("disc", False):
        kw["disc"] = D.band_mask(w, h, 90, 150)
    kw.update(v)
    lb.lqrhip_set_update_mode({"auto": -1, "levels": 5, "generic": 3}[path])
    if path == "levels":
        lb.lqrhip_set_band_levels(5)
    try:
        oracle.lqrx_set_debug(1); engine.lqrx_set_debug(1)
        ca, _ = H.init_carver(oracle, img, w - 30, h, **kw); cb, _ = H.init_carver(engine, img, w - 30, h, **kw)
        lb.lqrhip_prof_reset(); lb.lqrhip_prof_enable(1)
        assert ca.resize(w - 30, h) == L.LQR_OK and cb.resize(w - 30, h) == L.LQR_OK
        lb.lqrhip_prof_enable(0)
        if path == "auto":
            assert prof_launches(lb, "dp_update_tiled") > 0 and prof_launches(lb, "band_update") == 0, "delta_x %d fell to the one-wave kernels" % delta
        if path == "levels":
            assert prof_launches(lb, "band_levels") > 0
        (ea, ma, da), (eb, mb, db) = ca.debug_snapshot(), cb.debug_snapshot()
        assert np.array_equal(ea.view(np.int32), eb.view(np.int32)), "energies"
        assert np.array_equal(ma.view(np.int32), mb.view(np.int32)), "cumulative minima"
        assert np.array_equal(da[1:], db[1:]), "back pointers"
        assert np.array_equal(ca.vmap_dump()["data"], cb.vmap_dump()["data"]) and np.array_equal(ca.read_image(), cb.read_image())
        ca.destroy(); cb.destroy()
    finally:
        oracle.lqrx_set_debug(0); engine.lqrx_set_debug(0)
        lb.lqrhip_prof_enable(0); lb.lqrhip_set_update_mode(-1); lb.lqrhip_set_band_levels(-1)
    kw.pop("switch_freq")
    H.assert_same(H.run_case(oracle, img, w - 25, h - 17, **kw), H.run_case(engine, img, w - 25, h - 17, **kw), "delta %d %s both directions" % (delta, variant))


@pytest.mark.parametrize("delta", [5, 8, 10])
def test_wide_delta_group_of_nine_runs_the_full_width_tiled_kernels(oracle, engine, lib, delta):
    """a lock-step group with delta_x 5 .. 10 stays on k_dp_tile_p's general instantiations (the band is the whole width after a few hundred
    rows: the level kernel's images would stop at a collision and fall to the one-workgroup sweep), carved group after group if it is
    larger than the persistent grid holds; the backtrack is the parallel one whatever the group size"""
    w, h, n = 640, 300, 9
    imgs = [D.photo_like(w, h, 900 + i) if i % 2 else D.noise(w, h, 900 + i) for i in range(n)]
    rig = 2.0 if delta == 8 else 0.0
    cs = [L.Carver(engine, im, delta_x=delta, rigidity=rig).configure() for im in imgs]
    lib.lqrhip_prof_reset(); lib.lqrhip_prof_enable(1)
    assert L.resize_batch(engine, cs, w - 40, h - 12) == L.LQR_OK
    lib.lqrhip_prof_enable(0)
    assert prof_launches(lib, "dp_update_tiled") > 0 and prof_launches(lib, "band_update") == 0 and prof_launches(lib, "band_levels") == 0
    for c, im in zip(cs, imgs):
        ref = H.run_case(oracle, im, w - 40, h - 12, delta_x=delta, rigidity=rig)
        assert np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"]) and np.array_equal(c.read_image(), ref["image"])
    for c in cs:
        c.destroy()


def test_wide_delta_4k_rows_and_8k_columns(oracle, engine, lib):
    """tall and wide: 2300 rows at delta_x 10 = 767 blocks of 3 rows (the granule tags' block field), 6000 columns = 94 tiles"""
    for w, h, d in ((300, 2300, 10), (6000, 120, 7)):
        img = D.photo_like(w, h, 77)
        H.assert_same(H.run_case(oracle, img, w - 12, h, delta_x=d), H.run_case(engine, img, w - 12, h, delta_x=d), "%dx%d delta %d" % (w, h, d))


# ---------------------------------------------------------------- the 48-column-halo geometry of k_dp_tile_p (single images)
@pytest.mark.parametrize("w,h", [(31, 47), (33, 48), (64, 49), (97, 96), (129, 97), (700, 300), (1290, 145), (3000, 100)])
@pytest.mark.parametrize("px", [3, 2])
def test_wide_halo_tile_geometry_planes_and_seams(oracle, engine, lib, w, h, px):
    """geometry 3 of the persistent tiled kernels (32 own columns, 48-column halos that reach across the adjacent tile into the one behind
    it, 48-row blocks in two 24-row batches) against the oracle and against geometry 2: shapes around the tile (32), block (48) and
    batch (24) sizes; DP planes after 25 incremental updates with both tie rules and rigidity; whole resizes in both directions"""
    engine.lib.lqrhip_set_dp_persistent_px.argtypes = [ctypes.c_int]
    engine.lib.lqrhip_set_dp_persistent_px(px)
    img = D.photo_like(w, h, 500 + w) if w % 2 else D.noise(w, h, 500 + w)
    k = min(25, w - 3)
    try:
        for kw in (dict(switch_freq=0), dict(switch_freq=1000, rigidity=3.0)):
            oracle.lqrx_set_debug(1); engine.lqrx_set_debug(1)
            ca, _ = H.init_carver(oracle, img, w - k, h, **kw); cb, _ = H.init_carver(engine, img, w - k, h, **kw)
            assert ca.resize(w - k, h) == L.LQR_OK and cb.resize(w - k, h) == L.LQR_OK
            (ea, ma, da), (eb, mb, db) = ca.debug_snapshot(), cb.debug_snapshot()
            assert np.array_equal(ea.view(np.int32), eb.view(np.int32)) and np.array_equal(ma.view(np.int32), mb.view(np.int32)) and np.array_equal(da[1:], db[1:]), kw
            assert np.array_equal(ca.vmap_dump()["data"], cb.vmap_dump()["data"])
            ca.destroy(); cb.destroy()
        oracle.lqrx_set_debug(0); engine.lqrx_set_debug(0)
        H.assert_same(H.run_case(oracle, img, max(2, w - 9), max(2, h - 7)), H.run_case(engine, img, max(2, w - 9), max(2, h - 7)), "geometry %d, %dx%d" % (px, w, h))
    finally:
        oracle.lqrx_set_debug(0); engine.lqrx_set_debug(0)
        engine.lib.lqrhip_set_dp_persistent_px(0)


# ---------------------------------------------------------------- the carve and the energy update in one launch (k_carve_e)
@pytest.mark.parametrize("nrg", [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("ch", [1, 2, 3, 4])
def test_fused_carve_energy_planes_are_exact(oracle, engine, lib, nrg, ch):
    """k_carve_e (groups up to 4 images, delta_x <= 2): the wave that has moved a row refreshes that row's energies.  Every energy
    function x channel layout, with bias masks, delta_x 1 and 2: after 30 seams the energy plane (0 ULP), the cumulative minima and the
    back pointers equal the oracle's, with the fusion on and off; then whole resizes in both directions with attached layers"""
    engine.lib.lqrhip_set_carve_fused.argtypes = [ctypes.c_int]
    w, h = 210, 96
    img = (D.alpha_ramp if ch in (2, 4) else D.photo_like)(w, h, 70 + ch + nrg, channels=ch)
    try:
        for fused in (1, 0):
            engine.lib.lqrhip_set_carve_fused(fused)
            for kw in (dict(switch_freq=0), dict(switch_freq=0, delta_x=2, rigidity=2.0, pres=D.ellipse_mask(w, h), disc=D.band_mask(w, h, 40, 80))):
                kw = dict(kw, nrg_func=nrg)
                oracle.lqrx_set_debug(1); engine.lqrx_set_debug(1)
                ca, _ = H.init_carver(oracle, img, w - 30, h, **kw); cb, _ = H.init_carver(engine, img, w - 30, h, **kw)
                assert ca.resize(w - 30, h) == L.LQR_OK and cb.resize(w - 30, h) == L.LQR_OK
                (ea, ma, da), (eb, mb, db) = ca.debug_snapshot(), cb.debug_snapshot()
                assert np.array_equal(ea.view(np.int32), eb.view(np.int32)), ("energies", fused, kw.get("delta_x", 1))
                assert np.array_equal(ma.view(np.int32), mb.view(np.int32)) and np.array_equal(da[1:], db[1:]), ("DP planes", fused)
                ca.destroy(); cb.destroy()
            oracle.lqrx_set_debug(0); engine.lqrx_set_debug(0)
            kw = dict(nrg_func=nrg, pres=D.ellipse_mask(w, h), resize_aux_layers=True, output_seams=True)
            H.assert_same(H.run_case(oracle, img, w - 21, h - 13, **kw), H.run_case(engine, img, w - 21, h - 13, **kw), "fused %d nrg %d ch %d" % (fused, nrg, ch))
    finally:
        oracle.lqrx_set_debug(0); engine.lqrx_set_debug(0)
        engine.lib.lqrhip_set_carve_fused(1)


@pytest.mark.parametrize("n", [2, 4, 5])
def test_fused_carve_energy_groups(oracle, engine, lib, n):
    """groups of 2 and 4 run the fused kernel, 5 the two kernels: every image against the oracle; 70 seams so that the frozen planes
    are caught up (lag 32) inside the session, and the last seams go down to a width of 2"""
    w, h = 74, 140
    imgs = [D.noise(w, h, 40 + i) if i % 2 else D.photo_like(w, h, 40 + i) for i in range(n)]
    cs = [L.Carver(engine, im).configure() for im in imgs]
    assert L.resize_batch(engine, cs, 2, h) == L.LQR_OK
    for c, im in zip(cs, imgs):
        ref = H.run_case(oracle, im, 2, h)
        assert np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"]) and np.array_equal(c.read_image(), ref["image"])
    for c in cs:
        c.destroy()
