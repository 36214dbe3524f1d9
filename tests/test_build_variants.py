"""The engine's build carries one internal, unversioned LLVM option (-mllvm -amdgpu-sched-strategy=max-ilp: it only
reschedules instructions; Makefile).  The parity suite must be green with and without it: this test (-m gpu) compiles
the library a second time WITHOUT the option into tests/c/build/nosched-<hash of the sources>/ and runs a cross-section of the parity cases -- every
update_mmap form, both tie rules, rigidity, delta_x 2, masks, the tolerance boundary -- through that build."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

import datasets as D
import harness as H
import lqr_ctypes as L

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gimp-lqr-plugin_amd")
OUT = os.path.join(ROOT, "tests", "c", "build")


def source_hash():
    """sha1 over everything the library is built from: a second build is only ever reused for exactly these sources (round 5
    shipped a prebuilt variant four hours older than the kernels; whether `make` rebuilds depends on mtimes that a copy of
    the tree need not preserve)"""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(PKG, "csrc", "*")) + glob.glob(os.path.join(PKG, "host", "*")) + glob.glob(os.path.join(ROOT, "include", "*")) + [os.path.join(PKG, "Makefile")]):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


@pytest.fixture(scope="module")
def plain_build():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this machine")
    key = source_hash()
    out = os.path.join(OUT, "nosched-" + key)
    for old in os.listdir(OUT) if os.path.isdir(OUT) else []:          # variants of other source states (and round 5's unkeyed one)
        if (old.startswith("nosched") and old != "nosched-" + key) or old == "liblqr-hip-default-sched.so":
            p = os.path.join(OUT, old)
            shutil.rmtree(p) if os.path.isdir(p) else os.remove(p)
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "liblqr-hip-default-sched.so")
    # the package's own Makefile with the option switched off (SCHED=), objects and library in a directory keyed by the sources' hash
    subprocess.check_call(["make", "-C", PKG, "-j8", "SCHED=", "BUILD=" + os.path.join(out, "obj"), "OUT=" + so], stdout=subprocess.DEVNULL)
    return L.Api(so, "")


CASES = [
    ("photo 300x160 -> 260x140", lambda: D.photo_like(300, 160, 73), 260, 140, {}),
    ("noise 1200x200 -> 1150x200, every seam switches side", lambda: D.noise(1200, 200, 5), 1150, 200, dict(switch_freq=1000)),
    ("masks + rigidity + delta 2", lambda: D.photo_like(420, 260, 77), 380, 240,
     dict(pres=D.ellipse_mask(420, 260), disc=D.band_mask(420, 260, 60, 110), rigmask=D.top_half_mask(420, 260), rigidity=6.0, delta_x=2)),
    ("flat blocks, null energy", lambda: D.flat_blocks(276, 80, 4), 216, 80, dict(nrg_func=L.LQR_EF_NULL, pres=D.ellipse_mask(276, 80))),
    ("enlarge", lambda: D.photo_like(120, 90, 9), 170, 90, {}),
]


@pytest.mark.parametrize("mode", [-1, 0, 2, 3])
def test_parity_without_the_scheduler_option(plain_build, oracle, mode):
    plain_build.lib.lqrhip_set_update_mode.argtypes = [ctypes.c_int]
    plain_build.lib.lqrhip_set_update_mode(mode)
    try:
        for what, make, nw, nh, kw in CASES:
            img = make()
            ref = H.run_case(oracle, img, nw, nh, **kw)
            try:
                H.assert_same(ref, H.run_case(plain_build, img, nw, nh, **kw), "%s, update mode %d" % (what, mode))
            except AssertionError as ex:
                # (round 5: this test failed ONCE in a full-suite run and never again -- DESIGN.md 8.  Should it happen again: is the
                # mismatch a state of the process, or a passing event?  The same case three more times, and the engine's last error.)
                again = []
                for _ in range(3):
                    try:
                        H.assert_same(ref, H.run_case(plain_build, img, nw, nh, **kw), "again")
                        again.append("ok")
                    except AssertionError as ex2:
                        again.append(str(ex2)[:120])
                raise AssertionError("%s\n  the same case three more times: %s\n  last error: %r" % (ex, again, plain_build.lib.lqrhip_last_error()))
    finally:
        plain_build.lib.lqrhip_set_update_mode(-1)


def test_tolerance_boundary_without_the_scheduler_option(plain_build, oracle):
    import test_tolerance_boundary as TB
    import tolerance_case as T
    cols, b = TB.seam_columns(plain_build)
    _, a = TB.seam_columns(oracle)
    assert cols == T.EXPECTED_SEAM2_COLUMNS
    H.assert_same(a, b, "tolerance boundary, default scheduler")
