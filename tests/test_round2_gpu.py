"""Round-2 parity cases (-m gpu): the kernel variants and host paths round 1's suite did not reach.

  * k_dp_tile, the full DP of batches whose tiles are not co-resident (the 64 x 4K bench workload): forced
    through lqrhip_set_dp_persistent_limit(0) on small cases with both tie rules, and reached naturally by a
    26 x 3840x48 batch and by the 64 x 4K batch itself;
  * BASELINE config 4 as stated: 64 4K images in one lock-step batch;
  * delta_x > 2 with rigidity 0 (seams wander: k_emap_update's sample window), ADVICE round 1;
  * batches that mix carvers with and without second DP planes; masks on a transposed carver;
  * lqrx_carver_reload_device_batch, lqrx_carver_read_image_device + dist.gather (the bench's data path);
  * the seam-map colour ramp on the device.
"""
import ctypes
import os

import numpy as np
import pytest

import datasets as D
import harness as H
import lqr_ctypes as L
from test_fullsize_gpu import check_vertical_properties

pytestmark = pytest.mark.gpu


@pytest.fixture
def no_persistent_dp(engine):
    engine.lib.lqrhip_set_dp_persistent_limit.argtypes = [ctypes.c_int]
    engine.lib.lqrhip_set_dp_persistent_limit(0)
    yield
    engine.lib.lqrhip_set_dp_persistent_limit(-1)


def both(oracle, engine, img, nw, nh, **kw):
    a = H.run_case(oracle, img, nw, nh, **kw)
    b = H.run_case(engine, img, nw, nh, **kw)
    H.assert_same(a, b, "%sx%s->%sx%s %s" % (img.shape[1], img.shape[0], nw, nh, sorted(kw)))
    return a, b


@pytest.mark.parametrize("freq", [2, 1000])
def test_dp_tile_forced_both_tie_rules(oracle, engine, no_persistent_dp, freq):
    """every full DP through k_dp_tile: switch_freq 2 rebuilds with leftright = 1 then 0 again, 1000 rebuilds
    after every seam (k_dp_tile on every width, both tie rules); rigidity exercises the RIG instantiations"""
    both(oracle, engine, D.photo_like(700, 150, 201), 640, 150, switch_freq=freq)
    both(oracle, engine, D.flat_blocks(450, 97, 202), 400, 80, switch_freq=freq, rigidity=5.0)
    both(oracle, engine, D.noise(193, 33, 203), 150, 33, switch_freq=freq)          # one tile + a sliver, one row block + 1 row
    both(oracle, engine, D.noise(385, 64, 204, channels=1), 300, 64, switch_freq=freq, nrg_func=0)


def test_dp_tile_natural_batch_26x3840x48(oracle, engine):
    """30 tiles x 26 images = 780 workgroups > 3 per CU: launch_dp picks k_dp_tile by itself"""
    imgs = [D.noise(3840, 48, 300 + i) for i in range(26)]
    cs = [L.Carver(engine, im).configure() for im in imgs]
    assert L.resize_batch(engine, cs, 3800, 48) == L.LQR_OK
    for i in (0, 13, 25):
        ref = H.run_case(oracle, imgs[i], 3800, 48)
        assert np.array_equal(cs[i].vmap_dump()["data"], ref["vmap"]["data"]), i
        assert np.array_equal(cs[i].read_image(), ref["image"]), i
    for c in cs:
        c.destroy()


def test_config4_batch_of_64_4k_images(oracle, engine):
    """BASELINE config 4 as stated: 64 independent 4K RGBA images, 200 seams each, one lock-step batch
    (k_band_update_tw + k_dp_tile).  Images 0 and 63 are compared with the oracle bit for bit; every image must
    satisfy the removal identity (output = input minus the seam pixels), three of them the full seam properties."""
    n = 64
    imgs = [D.noise(3840, 2160, 100 + i) for i in range(n)]
    cs = [L.Carver(engine, im).configure() for im in imgs]
    assert L.resize_batch(engine, cs, 3640, 2160) == L.LQR_OK
    for i in (0, 63):
        ref = H.run_case(oracle, imgs[i], 3640, 2160)
        assert np.array_equal(cs[i].vmap_dump()["data"], ref["vmap"]["data"]), i
        assert np.array_equal(cs[i].read_image(), ref["image"]), i
    for i in range(1, 63):
        vm = cs[i].vmap_dump()["data"]
        out = cs[i].read_image()
        assert ((vm > 0).sum(axis=1) == 200).all(), i
        assert np.array_equal(out.reshape(-1, 4), imgs[i][vm == 0]), i
        if i in (1, 31, 62):
            check_vertical_properties(imgs[i], out, vm, 200, sample=4)
    for c in cs:
        c.destroy()


@pytest.mark.parametrize("delta", [3, 5, 10, 16])
@pytest.mark.parametrize("dataset", ["noise", "photo_like"])
def test_large_delta_without_rigidity(oracle, engine, delta, dataset):
    """seams that move delta_x per row: the energy update must stage up to 4*delta_x + 2 samples per row"""
    img = D.DATASETS[dataset](220, 140, 400 + delta)
    both(oracle, engine, img, 170, 140, delta_x=delta, rigidity=0.0)
    both(oracle, engine, img, 190, 110, delta_x=delta, rigidity=0.0, nrg_func=0)


def test_batch_mixing_carvers_with_and_without_second_planes(oracle, engine):
    """carver A goes through a tiled update on its own (allocates m2 / least2), comes back to its original size and
    is flattened; B and C are fresh.  The three then run as one batch."""
    imgs = [D.photo_like(260, 120, 500 + i) for i in range(3)]
    out = []
    for api in (oracle, engine):
        cs = [L.Carver(api, im).configure(switch_freq=0) for im in imgs]
        assert cs[0].resize(240, 120) == L.LQR_OK
        assert cs[0].resize(260, 120) == L.LQR_OK
        assert cs[0].flatten() == L.LQR_OK
        if api is engine:
            assert L.resize_batch(api, cs, 230, 120) == L.LQR_OK
        else:
            for c in cs:
                assert c.resize(230, 120) == L.LQR_OK
        out.append([(c.read_image(), c.vmap_dump()["data"]) for c in cs])
        for c in cs:
            c.destroy()
    for (ia, va), (ib, vb) in zip(*out):
        assert np.array_equal(ia, ib) and np.array_equal(va, vb)


def test_masks_added_to_a_transposed_carver_with_offsets(oracle, engine):
    """a height-first resize leaves the carver transposed; masks with non-zero offsets are then added in image
    coordinates and a second resize uses them"""
    img = D.photo_like(150, 110, 600)
    out = []
    for api in (oracle, engine):
        c = L.Carver(api, img, delta_x=1, rigidity=3.0).configure(res_order=L.LQR_RES_ORDER_VERT)
        assert c.resize(150, 90) == L.LQR_OK
        assert c.getters()["orientation"] == 1
        assert c.bias_add(D.ellipse_mask(70, 50), 800, x_off=40, y_off=-12) == L.LQR_OK
        assert c.bias_add(D.band_mask(200, 30, 20, 90, channels=3), -600, x_off=-25, y_off=50) == L.LQR_OK
        assert c.rigmask_add(D.top_half_mask(60, 120, channels=2), x_off=100, y_off=-5) == L.LQR_OK
        assert c.resize(120, 70) == L.LQR_OK
        out.append((c.read_image(), c.vmap_dump(), c.getters()))
        c.destroy()
    assert out[0][2] == out[1][2]
    assert np.array_equal(out[0][1]["data"], out[1][1]["data"])
    assert np.array_equal(out[0][0], out[1][0])


def test_reload_from_device_memory_equals_a_fresh_carver(oracle, engine):
    """lqrx_carver_reload_device_batch (bench.py's per-step input hand-over): after a bidirectional resize with
    masks, reloading new pixels must give exactly what a fresh carver gives"""
    torch = pytest.importorskip("torch")
    w, h = 300, 180
    first = [D.photo_like(w, h, 700 + i) for i in range(3)]
    second = [D.noise(w, h, 710 + i) for i in range(3)]
    dev = torch.stack([torch.from_numpy(x) for x in second]).cuda()
    cs = [L.Carver(engine, im).configure() for im in first]
    assert cs[0].bias_add(D.ellipse_mask(w, h), 500) == L.LQR_OK          # makes carver 0 differ: masks must be dropped by the reload
    assert cs[0].resize(260, 150) == L.LQR_OK
    for c in cs[1:]:
        assert c.resize(270, 180) == L.LQR_OK
    assert L.reload_device_batch(engine, cs, [dev[i].data_ptr() for i in range(3)]) == L.LQR_OK
    for c in cs:
        g = c.getters()
        assert (g["width"], g["height"], g["orientation"], g["depth"]) == (w, h, 0, 0)
    assert L.resize_batch(engine, cs, 250, 160) == L.LQR_OK
    for im, c in zip(second, cs):
        ref = H.run_case(oracle, im, 250, 160)
        assert np.array_equal(c.read_image(), ref["image"])
        assert np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"])
        c.destroy()
    # refused on carvers with attached carvers
    c = L.Carver(engine, first[0]).configure()
    c.attach(D.ellipse_mask(w, h))
    assert L.reload_device_batch(engine, [c], [dev[0].data_ptr()]) == L.LQR_ERROR
    c.destroy()


def test_engine_shard_read_device_and_gather(oracle, engine):
    """bench.py's multi-GPU data path with the ENGINE as backend: image i -> rank i mod N (shard_indices), carve as one
    batch, lqrx_carver_read_image_device straight into a torch tensor, one dist.gather over RCCL (world size 1 here;
    the world-size-2 logic runs on gloo in tests/test_sharding.py)"""
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge._import_package()
    n_images, w, h, nw = 5, 200, 96, 170
    imgs = [D.photo_like(w, h, 800 + i) for i in range(n_images)]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    own_group = not dist.is_initialized()
    if own_group:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        mine = pkg.shard_indices(n_images, dist.get_rank(), dist.get_world_size())
        assert mine == list(range(n_images))
        cs = [L.Carver(engine, imgs[i]).configure() for i in mine]
        assert L.resize_batch(engine, cs, nw, h) == L.LQR_OK
        outs = torch.empty((len(mine), h, nw, 4), dtype=torch.uint8, device="cuda")
        for j, c in enumerate(cs):
            assert engine.lqrx_carver_read_image_device(c.p, outs[j].data_ptr()) == L.LQR_OK
        gathered = [torch.empty_like(outs)]
        dist.gather(outs, gathered, dst=0)
        torch.cuda.synchronize()
        got = gathered[0].cpu().numpy()
        for j, i in enumerate(mine):
            ref = H.run_case(oracle, imgs[i], nw, h)
            assert np.array_equal(got[j], ref["image"]), i
        for c in cs:
            c.destroy()
    finally:
        if own_group:
            dist.destroy_process_group()


def test_vmap_colour_ramp_on_the_device(oracle, engine):
    """write_vmap_to_layer's colour ramp (src/io_functions.c:249-279) over the dumped seam maps of
    config 3 (bidirectional) and config 5 (masks) at reduced scale; engine kernel vs the oracle's restatement"""
    cases = [(D.photo_like(480, 270, 3), 420, 210, {}),
             (D.photo_like(480, 270, 5), 420, 270, dict(pres=D.ellipse_mask(480, 270), disc=D.band_mask(480, 270, 90, 130), rigidity=10.0))]
    colours = [((1.0, 1.0, 0.0), (0.2, 0.0, 0.0)),          # the plug-in's defaults: yellow -> dark red (main.c)
               ((0.123456789, 0.999, 1 / 3), (0.7071067811865476, 0.0, 1.0))]
    for img, nw, nh, kw in cases:
        for api in (oracle, engine):
            c, _ = H.init_carver(api, img, nw, nh, output_seams=True, **kw)
            assert c.resize(nw, nh) == L.LQR_OK
            res = []

            def cb(v, _, res=res, api=api):
                for cs_, ce_ in colours:
                    res.append(L.vmap_to_rgba(api, v, cs_, ce_))
                return L.LQR_OK
            fn = L.VMAP_FUNC(cb)
            assert api.lqr_vmap_list_foreach(api.lqr_vmap_list_start(c.p), fn, None) == L.LQR_OK
            c.destroy()
            if api is oracle:
                want = res
            else:
                assert len(res) == len(want) and len(res) >= 2
                for a, b in zip(want, res):
                    assert a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.parametrize("where", ["left", "right", "both"])
def test_both_carve_sides_and_origin(oracle, engine, where):
    """the carve moves the shorter side of the seam: a discard band at the left edge sends every seam there (the left
    part moves right, the planes' origin advances seam by seam), one at the right edge keeps the origin at 0, two
    bands alternate.  Seam maps, pixels AND the DP planes after the last incremental update must match the oracle
    (lqrx debug snapshot reads the planes through the origin)."""
    w, h = 420, 150
    img = D.photo_like(w, h, 900)
    disc = np.zeros((h, w, 4), np.uint8)
    if where in ("left", "both"):
        disc[:, 6:60] = 255
    if where in ("right", "both"):
        disc[:, w - 70:w - 9] = 255
    for mode in (-1, 0, 2):
        engine.lib.lqrhip_set_update_mode.argtypes = [ctypes.c_int]
        engine.lib.lqrhip_set_update_mode(mode)
        try:
            both(oracle, engine, img, w - 90, h - 20, disc=disc, output_seams=True)
            snaps = []
            for api in (oracle, engine):
                api.lqrx_set_debug(1)
                c, _ = H.init_carver(api, img, w - 45, h, disc=disc, switch_freq=0)
                assert c.resize(w - 45, h) == L.LQR_OK
                snaps.append(c.debug_snapshot())
                api.lqrx_set_debug(0)
                c.destroy()
            (ea, ma, da), (eb, mb, db) = snaps
            assert np.array_equal(ea, eb) and np.array_equal(ma, mb) and np.array_equal(da[1:], db[1:])
        finally:
            engine.lib.lqrhip_set_update_mode(-1)


@pytest.mark.parametrize("nsub", [2, 4])
def test_sub_batches_on_separate_streams(oracle, engine, nsub):
    """a lock-step group split over several HIP streams (opt-in, bench.py --sub-batches): every image must still come
    out as the oracle's, whatever the interleaving; persistent (co-residency-dependent) kernels are never chosen for
    such batches"""
    engine.lib.lqrhip_set_sub_batches.argtypes = [ctypes.c_int]
    imgs = [D.photo_like(900, 260, 950 + i) if i % 2 else D.noise(900, 260, 950 + i) for i in range(9)]
    engine.lib.lqrhip_set_sub_batches(nsub)
    try:
        cs = [L.Carver(engine, im).configure() for im in imgs]
        assert L.resize_batch(engine, cs, 840, 230) == L.LQR_OK           # both directions: transposes and flattens per sub-batch
    finally:
        engine.lib.lqrhip_set_sub_batches(0)            # back to the engine's own choice
    for im, c in zip(imgs, cs):
        ref = H.run_case(oracle, im, 840, 230)
        assert np.array_equal(c.read_image(), ref["image"])
        assert np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"])
        c.destroy()


@pytest.mark.parametrize("px", [2, 3, 4])
def test_persistent_sweep_pixels_per_lane(oracle, engine, px):
    """k_dp_tile_p runs with 2 pixels per lane (128-column tiles) while twice the tiles fit the residency bound and with
    4 (256-column tiles) beyond that; round 6: geometry 3 = 2 pixels per lane, 32 own columns + 48-column halos, 48-row blocks
    (single images); all pinned here on the same cases: full builds with both tie rules, incremental
    updates, rigidity, tiles that straddle the image borders, a batch"""
    engine.lib.lqrhip_set_dp_persistent_px.argtypes = [ctypes.c_int]
    engine.lib.lqrhip_set_dp_persistent_px(px)
    try:
        both(oracle, engine, D.photo_like(700, 300, 990), 640, 260, output_seams=True)
        both(oracle, engine, D.noise(450, 131, 991), 400, 131, switch_freq=1000, rigidity=3.0)
        both(oracle, engine, D.flat_blocks(129, 70, 992), 100, 70)
        both(oracle, engine, D.noise(63, 40, 993, channels=1), 40, 40, nrg_func=0)
        imgs = [D.photo_like(520, 200, 994 + i) for i in range(3)]
        cs = [L.Carver(engine, im).configure() for im in imgs]
        assert L.resize_batch(engine, cs, 470, 200) == L.LQR_OK
        for im, c in zip(imgs, cs):
            ref = H.run_case(oracle, im, 470, 200)
            assert np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"])
            c.destroy()
    finally:
        engine.lib.lqrhip_set_dp_persistent_px(0)


def test_batch_with_attached_carvers_enlarges_in_steps(oracle, engine):
    """a lock-step batch whose carvers each carry masks AND attached carvers (resize_aux_layers), enlarged past enl_step
    so that every session ends in E14 (inflate) with several jobs per image in its one launch; both directions.
    Each image must come out as the oracle's one-by-one run: pixels, attached carvers, final seam map."""
    w, h = 150, 90
    imgs = [D.photo_like(w, h, 990 + i) if i % 2 else D.noise(w, h, 990 + i) for i in range(4)]
    pres, disc, rig = D.ellipse_mask(w, h), D.band_mask(w, h, 20, 50), D.top_half_mask(w, h)
    kw = dict(pres=pres, disc=disc, rigmask=rig, rigidity=2.0, resize_aux_layers=True, no_disc_on_enlarge=False, enl_step=130)
    nw, nh = 240, 100                                   # 150 -> 194 -> 240 in two sessions, then 90 -> 100
    cs = [H.init_carver(engine, im, nw, nh, **kw)[0] for im in imgs]
    assert L.resize_batch(engine, cs, nw, nh) == L.LQR_OK
    for im, c in zip(imgs, cs):
        ref = H.run_case(oracle, im, nw, nh, **kw)
        got, _ = c.read_scanlines()
        assert np.array_equal(got, ref["image"])
        assert len(c.aux) == len(ref["aux"]) == 3
        for a, b in zip(c.aux, ref["aux"]):
            assert np.array_equal(a.read_scanlines()[0], b)
        assert np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"])
        c.destroy()


@pytest.mark.parametrize("kind,w,h,n_seams", [("photo", 1500, 700, 60), ("noise", 2000, 333, 50), ("noise", 1024, 1201, 30),
                                              ("photo", 1300, 66, 30), ("noise", 3000, 97, 25)])
def test_dp_planes_on_images_wider_than_the_band_window(oracle, engine, kind, w, h, n_seams):
    """images wider than the band kernel's 896-column window, so that the window follows the seam, slots straddle the
    changes and (lane staging) only part of a slot's lanes is loaded and stored: en, m and the back
    pointers after the last incremental update must equal the oracle's bit for bit, for every form of update_mmap"""
    img = D.photo_like(w, h, w + h) if kind == "photo" else D.noise(w, h, w + h)
    engine.lib.lqrhip_set_update_mode.argtypes = [ctypes.c_int]
    oracle.lqrx_set_debug(1)
    c, _ = H.init_carver(oracle, img, w - n_seams, h, switch_freq=0)
    assert c.resize(w - n_seams, h) == L.LQR_OK
    ea, ma, da = c.debug_snapshot()
    oracle.lqrx_set_debug(0)
    c.destroy()
    try:
        for mode in (-1, 0, 2):
            engine.lib.lqrhip_set_update_mode(mode)
            engine.lqrx_set_debug(1)
            c, _ = H.init_carver(engine, img, w - n_seams, h, switch_freq=0)
            assert c.resize(w - n_seams, h) == L.LQR_OK
            eb, mb, db = c.debug_snapshot()
            engine.lqrx_set_debug(0)
            c.destroy()
            assert np.array_equal(ea, eb) and np.array_equal(ma, mb) and np.array_equal(da[1:], db[1:]), (kind, w, h, mode)
    finally:
        engine.lqrx_set_debug(0)
        engine.lib.lqrhip_set_update_mode(-1)
