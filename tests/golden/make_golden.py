"""Generates tests/golden/*.npz: small inputs + the outputs the CPU oracle produces
for them through the C ABI.  These are the ORACLE's outputs (regression protection); the same 17
cases run through the genuine liblqr are tests/golden/ref/fixtures_*.npz (scripts/ref_engine/) -- identical
on 16, the 17th is spec delta 6.  These fixtures freeze the oracle's behaviour so
that (a) an oracle regression is caught on CPU and (b) the GPU engine is checked
against committed data as well as against the live oracle.

    python tests/golden/make_golden.py        # rewrites the fixtures
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import datasets as D      # noqa: E402
import harness as H       # noqa: E402
import lqr_ctypes as L    # noqa: E402


def cases():
    """name -> (img, new_w, new_h, kwargs): every array below is stored in the fixture"""
    c = {}
    c["noise_64x48_shrink"] = (D.noise(64, 48, 1), 50, 48, {})
    c["photo_64x48_bidir"] = (D.photo_like(64, 48, 2), 52, 40, {})
    c["photo_64x48_vert_first"] = (D.photo_like(64, 48, 3), 52, 40, dict(res_order=L.LQR_RES_ORDER_VERT))
    c["flat_48x32_ties"] = (D.flat_blocks(48, 32, 4), 36, 32, {})
    c["alpha_56x40"] = (D.alpha_ramp(56, 40, 5), 44, 40, {})
    c["rgb_56x40_norm"] = (D.photo_like(56, 40, 6, channels=3), 46, 34, dict(nrg_func=L.LQR_EF_GRAD_NORM))
    c["grey_56x40_luma"] = (D.photo_like(56, 40, 7, channels=1), 46, 40, dict(nrg_func=L.LQR_EF_LUMA_GRAD_SUMABS))
    c["greya_40x40_sumabs"] = (D.alpha_ramp(40, 40, 8, channels=2), 30, 40, dict(nrg_func=L.LQR_EF_GRAD_SUMABS))
    c["rigid_d2_64x48"] = (D.photo_like(64, 48, 9), 50, 48, dict(rigidity=10.0, delta_x=2))
    c["enlarge_48x32"] = (D.photo_like(48, 32, 10), 60, 32, {})
    c["enlarge_multistep_48x32"] = (D.photo_like(48, 32, 11), 90, 32, {})
    c["lqrback_48x32"] = (D.photo_like(48, 32, 12), 38, 26, dict(scaleback=True))
    c["masks_64x48"] = (D.photo_like(64, 48, 13), 50, 40,
                        dict(pres=D.ellipse_mask(64, 48), disc=D.band_mask(64, 48, 6, 16), rigmask=D.top_half_mask(64, 48),
                             rigidity=5.0, resize_aux_layers=True, output_seams=True))
    c["null_energy_disc_40x24"] = (D.noise(40, 24, 14), 32, 24, dict(disc=D.band_mask(40, 24, 10, 18), nrg_func=L.LQR_EF_NULL))
    c["switch_every_seam_48x32"] = (D.photo_like(48, 32, 15), 30, 32, dict(switch_freq=1000))
    # heavily tied maps: the case in which update_mmap's band used to shrink past the carved pixel's children
    # (spec delta 6)
    c["null_energy_masks_276x80"] = (D.noise(276, 80, 790234955, channels=1), 260, 80,
                                     dict(pres=D.ellipse_mask(276, 80), disc=D.band_mask(276, 80, 55, 92), nrg_func=L.LQR_EF_NULL))
    # update_mmap's keep rule exactly at its boundary, |m_old - m_new| == 1e-5f (tests/tolerance_case.py; DESIGN.md
    # section 2, spec delta 4): the stale value is kept, seam 2 runs through it
    import tolerance_case as T
    timg, tmask = T.build()
    c["tolerance_boundary_2000x3"] = (timg, T.W - 2, T.H, dict(pres=tmask, pres_coeff=T.FACTOR, nrg_func=L.LQR_EF_NULL, switch_freq=0))
    return c


def main():
    api = L.oracle_api()
    only = set(sys.argv[1:])          # optional: names of the fixtures to (re)write
    for name, (img, nw, nh, kw) in cases().items():
        if only and name not in only:
            continue
        r = H.run_case(api, img, nw, nh, **kw)
        arrays = dict(img=img, new_size=np.array([nw, nh]), image=r["image"], vmap=r["vmap"]["data"],
                      vmap_meta=np.array([r["vmap"]["depth"], r["vmap"]["orientation"]]),
                      getters=np.array([r["getters"][k] for k in ("width", "height", "channels", "ref_width", "ref_height",
                                                                  "orientation", "depth")]))
        for k, v in kw.items():
            arrays["kw_" + k] = np.asarray(v)
        for i, a in enumerate(r["aux"]):
            arrays["aux%d" % i] = a
        for i, v in enumerate(r.get("vmaps", [])):
            arrays["dumped%d" % i] = v["data"]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
        print(name, img.shape, "->", r["image"].shape)


if __name__ == "__main__":
    main()
