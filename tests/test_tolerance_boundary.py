"""update_mmap's keep rule at its boundary: a constructed input where |m_old - m_new| is EXACTLY 1e-5f with an unchanged
parent (tests/tolerance_case.py; DESIGN.md section 2, spec delta 4).  The comparison is made in double
((double) fabsf(d) < 1e-5), so the stale value is kept and the second seam runs through it; a float comparison would
send the seam down column 1.  The oracle is checked on the CPU; every update_mmap kernel of the engine on the GPU."""
import ctypes

import numpy as np
import pytest

import harness as H
import lqr_ctypes as L
import tolerance_case as T


def seam_columns(api, **extra):
    img, mask = T.build()
    r = H.run_case(api, img, T.W - 2, T.H, pres=mask, pres_coeff=T.FACTOR, nrg_func=L.LQR_EF_NULL, switch_freq=0, **extra)
    vm = r["vmap"]["data"]
    assert r["vmap"]["depth"] == 2
    cols = [np.nonzero(vm[y])[0].tolist() for y in range(T.H)]
    assert all(c[0] == 0 for c in cols)                 # seam 1: column 0 on every row
    return tuple(c[1] for c in cols), r


def test_the_case_is_what_it_claims():
    ea, eb = T.energies()
    assert np.float32(ea - eb).view(np.uint32) == np.uint32(0x3727C5AC)      # 1e-5f, and the subtraction is exact
    assert float(np.float32(1e-5)) < 1e-5                                     # below the double constant: "kept"


def test_oracle_keeps_the_stale_value(oracle):
    cols, _ = seam_columns(oracle)
    assert cols == T.EXPECTED_SEAM2_COLUMNS


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["auto", "band", "band-mw", "generic", "levels"])
def test_engine_keeps_the_stale_value(oracle, engine, mode):
    """auto = tiled full-width update (k_dp_tile_p), band = k_band_update_tw, band-mw = k_band_update_mw, generic =
    k_band_update / k_dp_sweep (update mode 3: the kernels delta_x > 2 runs on)"""
    engine.lib.lqrhip_set_update_mode.argtypes = [ctypes.c_int]
    engine.lib.lqrhip_set_update_mode({"auto": -1, "band": 0, "band-mw": 2, "generic": 3, "levels": 5}[mode])
    extra = {}
    try:
        cols, b = seam_columns(engine, **extra)
        _, a = seam_columns(oracle, **extra)
    finally:
        engine.lib.lqrhip_set_update_mode(-1)
    assert cols == T.EXPECTED_SEAM2_COLUMNS
    H.assert_same(a, b, "tolerance boundary, " + mode)
