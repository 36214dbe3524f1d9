"""BASELINE.json's configs at FULL size on the GPU (-m gpu).  Where the oracle
finishes in seconds the comparison is direct (configs 2 and 4's per-image job);
the larger ones (config 3: 4K, 500+500 seams; config 5: 8K with masks, 1000 seams)
are checked through size-independent properties of seam carving:
  * every seam removes exactly one pixel per row and is delta_x-connected in the
    frame it was carved from;
  * the output equals the input with exactly the seam pixels removed (one-direction
    case) -- a checksum identity over the whole image;
  * a batch of B copies of one job gives B identical results, equal to the single job.
The direct comparison of configs 3 and 5 with the oracle at full size is in test_round3_gpu.py.
"""
import os

import numpy as np
import pytest

import datasets as D
import harness as H
import lqr_ctypes as L

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["auto", "band", "band-mw"], autouse=True)
def update_mode(request, engine):
    """Every test runs three times: with the engine's default choice of update_mmap kernel (the tiled
    full-width sweep at these sizes), with the band kernel the large batches use, and with the
    per-row-barrier band kernel k_band_update_mw (the default for rows wider than 4200 px)."""
    import ctypes
    lib = engine.lib
    lib.lqrhip_set_update_mode.argtypes = [ctypes.c_int]
    lib.lqrhip_set_update_mode({"auto": -1, "band": 0, "band-mw": 2}[request.param])
    yield request.param
    lib.lqrhip_set_update_mode(-1)


def seam_frame_positions(vm, k):
    """x position of seam k on every row, in the frame it was carved from"""
    h, w = vm.shape
    cols = np.argmax(vm == k, axis=1)
    alive = (vm == 0) | (vm > k)
    csum = np.cumsum(alive, axis=1)
    return csum[np.arange(h), cols] - 1, cols


def check_vertical_properties(img, out, vm, n_seams, delta=1, sample=12):
    h, w = vm.shape
    counts = (vm > 0).sum(axis=1)
    assert (counts == n_seams).all()
    assert vm.max() == n_seams
    # each level exactly once per row
    srt = np.sort(vm, axis=1)[:, -n_seams:]
    assert (srt == np.arange(1, n_seams + 1)[None, :]).all()
    for k in np.unique(np.linspace(1, n_seams, sample).astype(int)):
        xs, _ = seam_frame_positions(vm, k)
        assert np.abs(np.diff(xs)).max() <= delta, k
    keep = vm == 0
    assert np.array_equal(out.reshape(-1, out.shape[2]), img[keep])       # removal identity, pixel for pixel


def test_config2_fullhd_direct(oracle, engine):
    """config 2: 1920x1080 RGBA, 200 vertical seams -- oracle and engine bit-identical"""
    img = D.photo_like(1920, 1080, 2)
    a = H.run_case(oracle, img, 1720, 1080)
    b = H.run_case(engine, img, 1720, 1080)
    H.assert_same(a, b, "config2")
    check_vertical_properties(img, b["image"], b["vmap"]["data"], 200)


def test_config4_one_4k_image_direct_and_batch(oracle, engine):
    """config 4's per-image job (4K RGBA, 200 seams): direct comparison, then a lock-step
    batch of 3 images must reproduce the single-image results"""
    imgs = [D.noise(3840, 2160, 100 + i) for i in range(3)]
    ref = H.run_case(oracle, imgs[0], 3640, 2160)
    cs = [L.Carver(engine, im).configure() for im in imgs]
    assert L.resize_batch(engine, cs, 3640, 2160) == L.LQR_OK
    vm0 = cs[0].vmap_dump()["data"]
    assert np.array_equal(vm0, ref["vmap"]["data"])
    assert np.array_equal(cs[0].read_image(), ref["image"])
    for im, c in zip(imgs[1:], cs[1:]):
        check_vertical_properties(im, c.read_image(), c.vmap_dump()["data"], 200, sample=5)
    single = H.run_case(engine, imgs[2], 3640, 2160)
    assert np.array_equal(single["vmap"]["data"], cs[2].vmap_dump()["data"])
    for c in cs:
        c.destroy()


def test_config3_4k_bidirectional_properties(oracle, engine):
    """config 3: 3840x2160 -> 3340x1660, 500 vertical then 500 horizontal seams"""
    img = D.noise(3840, 2160, 3)
    c = L.Carver(engine, img).configure(dump_vmaps=True)
    assert c.resize(3340, 1660) == L.LQR_OK
    g = c.getters()
    assert (g["width"], g["height"], g["orientation"], g["depth"], g["ref_width"], g["ref_height"]) == \
        (3340, 1660, 1, 500, 3340, 2160)
    out = c.read_image()
    assert out.shape == (1660, 3340, 4)
    v1, v2 = c.dumped_vmaps()
    assert v1["orientation"] == 0 and v2["orientation"] == 1 and v1["depth"] == v2["depth"] == 500
    # width phase: the intermediate image is the input minus the 500 vertical seams
    mid = img[v1["data"] == 0].reshape(2160, 3340, 4)
    check_vertical_properties(img, mid, v1["data"], 500, sample=6)
    # height phase: same identity on the transposed problem
    vm2 = v2["data"]                                        # image orientation, 2160 x 3340
    assert ((vm2 > 0).sum(axis=0) == 500).all()
    midT = np.ascontiguousarray(mid.transpose(1, 0, 2))     # 3340 x 2160
    check_vertical_properties(midT, np.ascontiguousarray(out.transpose(1, 0, 2)), np.ascontiguousarray(vm2.T), 500, sample=6)
    c.destroy()


def test_config5_8k_masks_rigidity_properties(engine):
    """config 5: 7680x4320 RGBA, preservation ellipse (+1000), discard band (-1000),
    rigidity 10, delta_x 1, 1000 seams"""
    w, h, n = 7680, 4320, 1000
    img = D.noise(w, h, 5)
    pres = D.ellipse_mask(w, h)
    disc = D.band_mask(w, h, 1500, 2100)
    c, _ = H.init_carver(engine, img, w - n, h, pres=pres, disc=disc, rigidity=10.0)
    assert c.resize(w - n, h) == L.LQR_OK
    vm = c.vmap_dump()["data"]
    out = c.read_image()
    check_vertical_properties(img, out, vm, n, sample=4)
    # the masks did their job: seams concentrate in the discard band and avoid the ellipse
    # (the bias is coeff/2/w_start per pixel, a nudge rather than a wall on a noise image)
    removed = vm > 0
    inside = pres[:, :, 0] > 0
    overall = removed.mean()
    assert removed[:, 1500:2100].mean() > 2 * overall
    assert removed[inside].mean() < 0.5 * overall
    c.destroy()
