"""Round 6 (-m gpu): the parallel backtrack (k_vp_maps / k_vp_solve, csrc/k_backtrack.hip) -- the engine's choice for one or two images of
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
@pytest.mark.parametrize("delta", [1, 2, 3, 4])
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
