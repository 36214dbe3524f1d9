"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the
oracle): CPU run checks the oracle still reproduces them; the -m gpu run checks the
HIP engine reproduces them bit-exactly through the C ABI."""
import glob
import os

import numpy as np
import pytest

import harness as H
import lqr_ctypes as L

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
GETTERS = ("width", "height", "channels", "ref_width", "ref_height", "orientation", "depth")


def load_case(path):
    z = np.load(path)
    kw = {}
    for k in z.files:
        if k.startswith("kw_"):
            v = z[k]
            kw[k[3:]] = v if v.ndim else v.item()
    return z, kw


def check(api, path):
    z, kw = load_case(path)
    nw, nh = [int(v) for v in z["new_size"]]
    r = H.run_case(api, z["img"], nw, nh, **kw)
    assert np.array_equal(r["vmap"]["data"], z["vmap"]), "seam indices differ"
    assert [r["vmap"]["depth"], r["vmap"]["orientation"]] == list(z["vmap_meta"])
    assert np.array_equal(r["image"], z["image"]), "pixels differ"
    assert [r["getters"][k] for k in GETTERS] == list(z["getters"])
    for i, a in enumerate(r["aux"]):
        assert np.array_equal(a, z["aux%d" % i])
    for i, v in enumerate(r.get("vmaps", [])):
        assert np.array_equal(v["data"], z["dumped%d" % i])


def test_fixtures_exist():
    assert len(FIXTURES) >= 15


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_reproduces_golden(oracle, path):
    check(oracle, path)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_engine_reproduces_golden(engine, path):
    check(engine, path)
