"""Parity tests proper (-m gpu): the HIP engine vs the CPU oracle through the same C
ABI, on the same seeded inputs.  The bar (BASELINE.json north_star): seam indices
bit-exact, float energy within 1 ULP -- the engine is built to be bit-exact on
both, so the energy tolerance below is 0 ULP and the test says so.

Sizes are chosen so that the oracle finishes in seconds; BASELINE.json's full-size
configs are checked through size-independent properties in test_fullsize_gpu.py.
"""
import numpy as np
import pytest

import datasets as D
import harness as H
import lqr_ctypes as L

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["auto", "band", "band-mw", "levels"], autouse=True)
def update_mode(request, engine):
    """Every test runs four times: with the engine's default choice of update_mmap kernel (the tiled
    full-width sweep at these sizes), with the band kernel the large batches use, with the
    per-row-barrier band kernel k_band_update_mw (the default for rows wider than 4200 px), and with k_band_levels (round 5: the
    band on several compute units, tiles assigned level by level; round 4's k_band_tiles was removed in round 6)."""
    import ctypes
    lib = engine.lib
    lib.lqrhip_set_update_mode.argtypes = [ctypes.c_int]
    lib.lqrhip_set_update_mode({"auto": -1, "band": 0, "band-mw": 2, "levels": 5}[request.param])
    yield request.param
    lib.lqrhip_set_update_mode(-1)

ENERGY_TOLERANCE_ULP = 0      # north_star allows 1; the engine mirrors every rounding step


def ulp_diff(a, b):
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7fffffff), ai)
    bi = np.where(bi < 0, -(bi & 0x7fffffff), bi)
    return np.abs(ai - bi).max()


def both(oracle, engine, img, nw, nh, **kw):
    a = H.run_case(oracle, img, nw, nh, **kw)
    b = H.run_case(engine, img, nw, nh, **kw)
    H.assert_same(a, b, "%sx%s->%sx%s %s" % (img.shape[1], img.shape[0], nw, nh, sorted(kw)))
    return a, b


@pytest.mark.parametrize("nrg", range(7))
@pytest.mark.parametrize("ch", [1, 2, 3, 4])
def test_energy_map(oracle, engine, nrg, ch):
    img = D.alpha_ramp(75, 41, 10 + nrg, channels=ch) if ch in (2, 4) else D.photo_like(75, 41, 10 + nrg, channels=ch)
    ca = L.Carver(oracle, img).configure(nrg_func=nrg)
    cb = L.Carver(engine, img).configure(nrg_func=nrg)
    ea, eb = ca.energy(), cb.energy()
    assert ea.shape == eb.shape == (41, 75)
    assert ulp_diff(ea, eb) <= ENERGY_TOLERANCE_ULP
    ca.destroy(); cb.destroy()


def test_energy_with_bias(oracle, engine):
    img = D.photo_like(90, 50, 3)
    out = []
    for api in (oracle, engine):
        c = L.Carver(api, img)
        assert c.bias_add(D.ellipse_mask(90, 50), 1000) == L.LQR_OK
        assert c.bias_add(D.band_mask(60, 70, 5, 25), -700, x_off=-7, y_off=-9) == L.LQR_OK     # negative offsets, oversize
        assert c.bias_add(D.band_mask(30, 20, 2, 9, channels=3), 333, x_off=70, y_off=40) == L.LQR_OK   # clipped at the border
        out.append(c.energy())
        c.destroy()
    assert ulp_diff(out[0], out[1]) <= ENERGY_TOLERANCE_ULP


@pytest.mark.parametrize("dataset", ["noise", "photo_like", "flat_blocks", "alpha_ramp"])
def test_config1_shape_512(oracle, engine, dataset):
    """BASELINE config 1 (512x512 RGBA, 50 vertical seams, plug-in defaults) on every dataset"""
    img = D.DATASETS[dataset](512, 512, 1)
    both(oracle, engine, img, 462, 512)


def test_config2_scaled(oracle, engine):
    """BASELINE config 2 geometry at 1/2 scale: 960x540, 100 vertical seams"""
    both(oracle, engine, D.photo_like(960, 540, 2), 860, 540)


def test_config3_scaled_bidirectional(oracle, engine):
    """BASELINE config 3 at 1/4 scale: 960x540 -> 835x415 (125 + 125 seams, width first)"""
    both(oracle, engine, D.photo_like(960, 540, 3), 835, 415, output_seams=True, progress=True)


def test_config5_scaled_masks_rigidity(oracle, engine):
    """BASELINE config 5 at 1/8 scale: masks + rigidity + rigidity mask (rigidity x3), delta 1 and 2"""
    w, h = 960, 540
    pres = D.ellipse_mask(w, h)
    disc = D.band_mask(w, h, 187, 262)
    rig = D.top_half_mask(w, h)
    img = D.photo_like(w, h, 5)
    both(oracle, engine, img, w - 125, h, pres=pres, disc=disc, rigidity=10.0)
    both(oracle, engine, img, w - 60, h, pres=pres, disc=disc, rigmask=rig, rigidity=10.0, delta_x=2,
         resize_aux_layers=True, output_seams=True)


@pytest.mark.parametrize("ch", [1, 2, 3, 4])
def test_channel_layouts(oracle, engine, ch):
    img = D.alpha_ramp(131, 77, 20 + ch, channels=ch) if ch in (2, 4) else D.photo_like(131, 77, 20 + ch, channels=ch)
    both(oracle, engine, img, 101, 60)


@pytest.mark.parametrize("nrg", range(7))
def test_energy_functions(oracle, engine, nrg):
    both(oracle, engine, D.photo_like(150, 90, 30 + nrg), 120, 90, nrg_func=nrg)


@pytest.mark.parametrize("delta,rigidity", [(0, 0.0), (1, 4.0), (2, 0.0), (3, 25.0), (10, 1000.0)])
def test_delta_and_rigidity(oracle, engine, delta, rigidity):
    both(oracle, engine, D.photo_like(140, 100, 40 + delta), 110, 80, delta_x=delta, rigidity=rigidity)


@pytest.mark.parametrize("freq", [0, 1, 2, 3, 7, 1000])
def test_side_switch_schedules(oracle, engine, freq):
    """freq=1000 rebuilds the DP map after every seam; freq=0 never switches"""
    both(oracle, engine, D.flat_blocks(160, 90, 50 + freq), 110, 90, switch_freq=freq)


def test_tie_heavy_inputs(oracle, engine):
    flat = np.full((64, 200, 4), 128, np.uint8)
    both(oracle, engine, flat, 150, 50)
    stripes = np.zeros((80, 160, 3), np.uint8)
    stripes[:, ::8] = 255
    both(oracle, engine, stripes, 100, 80)
    both(oracle, engine, D.flat_blocks(300, 200, 9, nblocks=40), 200, 150)


def test_null_energy_with_masks(oracle, engine):
    """heavily tied maps (fuzz find): every pixel whose parent was carved must be recomputed"""
    w, h = 276, 80
    both(oracle, engine, D.noise(w, h, 790234955, channels=1), 260, 55, nrg_func=6,
         pres=D.ellipse_mask(w, h), disc=D.band_mask(w, h, w // 5, w // 3))
    both(oracle, engine, D.flat_blocks(300, 90, 5), 250, 90, nrg_func=6, pres=D.ellipse_mask(300, 90), delta_x=2)


def test_wide_bands_overflow_the_window(oracle, engine):
    """a band wider than the band kernel's window exercises the full-width continuation"""
    both(oracle, engine, D.noise(1500, 700, 61), 1440, 700)
    both(oracle, engine, D.photo_like(1800, 500, 62), 1740, 500)


def test_long_sessions_cross_the_frozen_plane_lag(oracle, engine):
    """more than 128 seams in one session: the frozen pixel/bias planes are compacted mid-session
    (FROZEN_LAG_MAX) and the energy update maps coordinates through up to 128 log entries"""
    both(oracle, engine, D.photo_like(900, 260, 63), 590, 260)
    both(oracle, engine, D.noise(700, 200, 64), 380, 150, pres=D.ellipse_mask(700, 200), disc=D.band_mask(700, 200, 90, 190))


def test_enlargement(oracle, engine):
    img = D.photo_like(120, 80, 70)
    both(oracle, engine, img, 150, 80)                     # one step
    both(oracle, engine, img, 260, 80)                     # several enl_step rounds with flatten in between
    both(oracle, engine, img, 150, 110, enl_step=120.0)
    both(oracle, engine, img, 100, 110)                    # shrink one way, enlarge the other
    both(oracle, engine, img, 150, 80, disc=D.band_mask(120, 80, 20, 40))     # no_disc_on_enlarge drops the mask


def test_lqr_back_and_masks_with_offsets(oracle, engine):
    img = D.photo_like(130, 90, 71)
    both(oracle, engine, img, 100, 70, scaleback=True)
    out = []
    for api in (oracle, engine):
        c = L.Carver(api, img, delta_x=1, rigidity=6.0)
        assert c.bias_add(D.ellipse_mask(60, 40), 900, x_off=50, y_off=-10) == L.LQR_OK
        assert c.rigmask_add(D.top_half_mask(200, 30, channels=2), x_off=-30, y_off=20) == L.LQR_OK
        c.configure()
        assert c.resize(105, 75) == L.LQR_OK
        out.append((c.read_image(), c.vmap_dump()["data"], c.getters()))
        c.destroy()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]


def test_interactive_sequence_persistent_carver(oracle, engine):
    """render_interactive (render.c:465-574): repeated resizes on one carver, inside and
    beyond the cached map, flatten, dump -- state must track the oracle call by call"""
    img = D.photo_like(140, 100, 72)
    cs = [L.Carver(api, img).configure() for api in (oracle, engine)]
    seq = [("r", 120, 100), ("r", 130, 100), ("r", 100, 100), ("r", 150, 100), ("r", 140, 90), ("f",), ("r", 120, 80),
           ("r", 140, 90), ("f",), ("f",), ("r", 139, 90), ("r", 170, 90)]
    for step in seq:
        rets = []
        for c in cs:
            rets.append(c.resize(step[1], step[2]) if step[0] == "r" else c.flatten())
        assert rets[0] == rets[1] == L.LQR_OK, step
        assert cs[0].getters() == cs[1].getters(), step
        assert np.array_equal(cs[0].read_image(), cs[1].read_image()), step
        va, vb = cs[0].vmap_dump(), cs[1].vmap_dump()
        assert va["depth"] == vb["depth"] and np.array_equal(va["data"], vb["data"]), step
    for c in cs:
        c.destroy()


def test_dp_state_after_incremental_updates(oracle, engine):
    """the DP planes themselves (en, m, back-pointers) after 40 incremental updates:
    this is where the 1e-5 keep-stale rule would show a divergence first"""
    img = D.photo_like(300, 160, 73)
    snaps = []
    for api in (oracle, engine):
        api.lqrx_set_debug(1)
        c = L.Carver(api, img).configure(switch_freq=0)
        assert c.resize(260, 160) == L.LQR_OK
        snaps.append(c.debug_snapshot())
        api.lqrx_set_debug(0)
        c.destroy()
    (ea, ma, da), (eb, mb, db) = snaps
    assert ea.shape == eb.shape == (160, 260)       # the carved frame after the 40th seam
    assert np.array_equal(ea, eb)
    assert np.array_equal(ma, mb)
    assert np.array_equal(da[1:], db[1:])


def test_batch_equals_one_by_one(oracle, engine):
    imgs = [D.photo_like(200, 120, 80 + i) for i in range(5)]
    cs = [L.Carver(engine, im).configure() for im in imgs]
    assert L.resize_batch(engine, cs, 160, 100) == L.LQR_OK
    for im, c in zip(imgs, cs):
        ref = H.run_case(oracle, im, 160, 100)
        assert np.array_equal(c.read_image(), ref["image"])
        assert np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"])
        c.destroy()
    # heterogeneous batch falls back to one-by-one and still matches
    cs = [L.Carver(engine, D.noise(60 + 10 * i, 40, i)).configure() for i in range(3)]
    assert L.resize_batch(engine, cs, 50, 40) == L.LQR_OK
    for i, c in enumerate(cs):
        ref = H.run_case(oracle, D.noise(60 + 10 * i, 40, i), 50, 40)
        assert np.array_equal(c.read_image(), ref["image"])
        c.destroy()


def test_edge_geometries(oracle, engine):
    both(oracle, engine, D.noise(2, 30, 1), 1, 30)          # down to a single column (finish_vsmap)
    both(oracle, engine, D.noise(30, 1, 2), 20, 1)          # single row
    both(oracle, engine, D.noise(3, 3, 3), 2, 2)
    both(oracle, engine, D.noise(40, 25, 4), 40, 25)        # no-op resize
    both(oracle, engine, D.noise(257, 65, 5), 200, 64)      # off-by-one around the 256-px chunking
    both(oracle, engine, D.noise(1031, 33, 6), 1000, 33)    # wider than one DP pass per thread


def test_error_returns_match(oracle, engine):
    for api in (oracle, engine):
        c = L.Carver(api, D.noise(8, 6, 19))
        assert c.resize(0, 6) == L.LQR_ERROR and c.resize(8, -1) == L.LQR_ERROR
        assert api.lqr_carver_set_enl_step(c.p, 1.0) == L.LQR_ERROR
        assert api.lqr_carver_init(c.p, 1, 0.0) == L.LQR_ERROR
        aux = L.Carver(api, D.noise(7, 6, 1), init=False)
        assert api.lqr_carver_attach(c.p, aux.p) == L.LQR_ERROR
        aux.destroy()
        c.destroy()


@pytest.mark.parametrize("ch", [1, 2, 3, 4])
def test_guess_new_size(oracle, engine, ch):
    """the plug-in's auto-size mask scan (src/layers_combo.c:275-392)"""
    rng = np.random.default_rng(ch)
    for (mw, mh, x_off, y_off, ow, oh) in [(640, 400, 0, 0, 640, 400), (500, 300, -37, 21, 640, 400), (700, 450, 50, -60, 640, 400),
                                           (64, 48, 600, 380, 640, 400), (10, 10, 700, 10, 640, 400)]:
        mask = rng.integers(0, 256, size=(mh, mw, ch), dtype=np.uint8)
        mask[mh // 4: mh // 2, mw // 3: 2 * mw // 3] = 255
        for direction in (0, 1):
            a = oracle.lqrx_guess_new_size(mask.ctypes.data, ch, mw, mh, x_off, y_off, ow, oh, direction)
            b = engine.lqrx_guess_new_size(mask.ctypes.data, ch, mw, mh, x_off, y_off, ow, oh, direction)
            assert a == b, (mw, mh, x_off, y_off, direction, a, b)
