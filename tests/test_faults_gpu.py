"""Round 6 (-m gpu): a resize must not hand LQR_ERROR -- or LQR_OK with a wrong map -- to a caller that only tests for LQR_NOMEM
(src/render.c:42-46,318, then :366 writes the carver to the user's layer).

The engine's fast kernels are spin protocols (k_dp_tile_p, k_band_levels) that end in a time-out when their workgroups are not
co-resident in time, and two structural self-checks watch every session (the seam log before the levels are committed; every
level exactly once per row, fused into the inflate pass).  Any of these ends the session in LQRHIP_EFAULT; host/lqr_carver.c
rolls the session back -- base layout and bookkeeping as before -- and carves it again on the kernels without spin waits.
lqrhip_debug_inject provokes each fault: 1 spin time-out, 2 failed activity prediction, 3 / 4 a seam-log entry out of the frame /
disconnected, 5 / 6 a committed level cleared / duplicated.

Checked here: with recovery (the default) every injected fault ends in LQR_OK and the genuine result, bit for bit; without it
(lqrhip_set_recovery(0)) in LQR_ERROR with the carver as it was before the failed session -- scan lines serve that image -- and
the NEXT resize of the same carver and of a fresh one is bit-exact; two faults in a row end in LQR_ERROR likewise."""
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
    lb.lqrhip_debug_inject.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lb.lqrhip_fault_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    for f in ("lqrhip_set_recovery", "lqrhip_set_no_spin", "lqrhip_set_selfcheck", "lqrhip_set_update_mode", "lqrhip_set_band_levels"):
        getattr(lb, f).argtypes = [ctypes.c_int]
    lb.lqrhip_get_no_spin.restype = ctypes.c_int
    lb.lqrhip_debug_fail_alloc.argtypes = [ctypes.c_int]
    yield lb
    lb.lqrhip_debug_fail_alloc(-1)
    lb.lqrhip_debug_inject(0, 0, 0); lb.lqrhip_set_recovery(1); lb.lqrhip_set_no_spin(0); lb.lqrhip_set_selfcheck(1)
    lb.lqrhip_set_update_mode(-1); lb.lqrhip_set_band_levels(-1)


def stats(lb, reset=False):
    st = (ctypes.c_ulonglong * 8)()
    assert lb.lqrhip_fault_stats(st, 1 if reset else 0) == 0
    return dict(zip(["timeouts", "predictions", "seamlog", "levels", "rolled_back", "injected", "redone", "_"], [int(x) for x in st]))


KINDS = {1: "timeouts", 2: "predictions", 3: "seamlog", 4: "seamlog", 5: "levels", 6: "levels"}


@pytest.mark.parametrize("kind", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("case", ["single", "single-both-directions", "masks-delta2", "enlarge-steps"])
def test_an_injected_fault_is_recovered_from_and_the_result_is_exact(oracle, engine, lib, kind, case):
    img, nw, nh, kw = {
        "single": (D.photo_like(300, 160, 73), 260, 160, {}),
        "single-both-directions": (D.photo_like(300, 160, 74), 262, 141, dict(output_seams=True)),
        "masks-delta2": (D.photo_like(420, 260, 77), 380, 240, dict(pres=D.ellipse_mask(420, 260), disc=D.band_mask(420, 260, 60, 110), rigmask=D.top_half_mask(420, 260),
                                                                  rigidity=6.0, delta_x=2, resize_aux_layers=True)),
        "enlarge-steps": (D.photo_like(120, 90, 9), 230, 90, dict(enl_step=140.0)),         # two enlargement sessions with a flatten in between
    }[case]
    ref = H.run_case(oracle, img, nw, nh, progress=True, **kw)
    stats(lib, reset=True)
    lib.lqrhip_debug_inject(kind, 7 if kind < 5 else 0, 1)                 # the fault falls into seam step 7 of the first session that has one (5 / 6: the first commit)
    got = H.run_case(engine, img, nw, nh, progress=True, **kw)
    s = stats(lib)
    lib.lqrhip_set_no_spin(0)
    assert s["injected"] == 1 and s[KINDS[kind]] >= 1 and s["rolled_back"] >= 1 and s["redone"] >= 1, s
    assert got["ret"] == L.LQR_OK
    H.assert_same(ref, got, "%s, injected fault %d" % (case, kind))      # incl. the progress events: nothing is reported twice


@pytest.mark.parametrize("kind", [1, 3, 5])
@pytest.mark.parametrize("n,mode", [(3, -1), (9, -1), (9, 0), (34, -1)])
def test_a_fault_in_a_lock_step_batch_is_recovered_from(oracle, engine, lib, kind, n, mode):
    """groups of 3 (full-width tiled kernels), 9 (k_band_levels; k_band_update_tw forced) and 34 (two sub-batch streams where the
    process has the queues): the fault is found by whichever sub-batch synchronises first, every sub-batch is rolled back and redone"""
    w, h = 520, 140
    imgs = [D.photo_like(w, h, 300 + i) if i % 2 else D.noise(w, h, 300 + i) for i in range(n)]
    lib.lqrhip_set_update_mode(mode)
    cs = [L.Carver(engine, im).configure() for im in imgs]
    stats(lib, reset=True)
    lib.lqrhip_debug_inject(kind, 11 if kind < 5 else 0, 1)
    assert L.resize_batch(engine, cs, w - 31, h - 9) == L.LQR_OK
    s = stats(lib)
    lib.lqrhip_set_no_spin(0)
    assert s["injected"] == 1 and s["rolled_back"] >= 1, s
    for c, im in zip(cs, imgs):
        ref = H.run_case(oracle, im, w - 31, h - 9)
        assert np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"])
        assert np.array_equal(c.read_image(), ref["image"])
    for c in cs:
        c.destroy()


@pytest.mark.parametrize("kind", [1, 4, 6])
@pytest.mark.parametrize("twice", [False, True])
def test_without_recovery_the_carver_is_as_before_and_the_next_resize_is_exact(oracle, engine, lib, kind, twice):
    """twice = False: recovery switched off; True: recovery on, but the redone session is hit too.  Either way LQR_ERROR, the carver
    serves the image it held before the failed session, and resizing it again gives the genuine result."""
    img = D.photo_like(300, 160, 81)
    if not twice:
        lib.lqrhip_set_recovery(0)
    c, _ = H.init_carver(engine, img, 260, 140)
    lib.lqrhip_debug_inject(kind, 5 if kind < 5 else 0, 2 if twice else 1)
    assert c.resize(260, 140) == L.LQR_ERROR
    lib.lqrhip_debug_inject(0, 0, 0); lib.lqrhip_set_no_spin(0); lib.lqrhip_set_recovery(1)
    g = c.getters()
    assert (g["width"], g["height"], g["depth"], g["orientation"]) == (300, 160, 0, 0), g      # the first session failed: nothing happened
    lines, nlines = c.read_scanlines()
    assert nlines == 160 and np.array_equal(lines, img)
    assert np.array_equal(c.vmap_dump()["data"], np.zeros((160, 300), np.int32))
    # the same carver again, and a fresh one
    ref = H.run_case(oracle, img, 260, 140)
    assert c.resize(260, 140) == L.LQR_OK
    assert np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"]) and np.array_equal(c.read_image(), ref["image"])
    c.destroy()
    H.assert_same(ref, H.run_case(engine, img, 260, 140), "a fresh carver after a failed resize")


def test_a_fault_in_the_second_direction_keeps_the_first(oracle, engine, lib):
    """LQR_ERROR from the second direction's session (recovery off): the carver holds the first direction's result -- a consistent
    multi-size image -- and the same resize, asked for again, finishes the job exactly"""
    img = D.photo_like(300, 160, 82)
    lib.lqrhip_set_recovery(0)
    c, _ = H.init_carver(engine, img, 260, 140)
    # seam-log indices are per session: index 15 exists in both directions (40 and 20 seams); hit the SECOND session only by
    # carving the first direction in a call of its own, as the plug-in's interactive mode does
    assert c.resize(260, 160) == L.LQR_OK
    lib.lqrhip_debug_inject(4, 15, 1)
    assert c.resize(260, 140) == L.LQR_ERROR
    lib.lqrhip_set_recovery(1)
    co, _ = H.init_carver(oracle, img, 260, 140)
    assert co.resize(260, 160) == L.LQR_OK
    lines, _ = c.read_scanlines()
    ref1, _ = co.read_scanlines()
    assert lines.shape == ref1.shape and np.array_equal(lines, ref1)
    assert c.resize(260, 140) == L.LQR_OK and co.resize(260, 140) == L.LQR_OK
    assert np.array_equal(c.read_image(), co.read_image()) and np.array_equal(c.vmap_dump()["data"], co.vmap_dump()["data"])
    c.destroy(); co.destroy()


def test_a_deeper_session_of_a_multi_size_image_is_recovered_from_the_base_layout(oracle, engine, lib):
    """an interactive-style carver (src/render.c:465-574): 300 -> 280 builds 20 levels; 280 -> 250 continues on the SAME working
    planes (max_level > 1).  A fault in that second session loses them: the redo lays them out again from the pixels of the base
    layout that carry no level yet (k_wk_init_visible)."""
    img = D.photo_like(300, 120, 83)
    ce, _ = H.init_carver(engine, img, 280, 120)
    co, _ = H.init_carver(oracle, img, 280, 120)
    assert ce.resize(280, 120) == L.LQR_OK and co.resize(280, 120) == L.LQR_OK
    stats(lib, reset=True)
    lib.lqrhip_debug_inject(1, 9, 1)
    assert ce.resize(250, 120) == L.LQR_OK and co.resize(250, 120) == L.LQR_OK
    s = stats(lib); lib.lqrhip_set_no_spin(0)
    assert s["rolled_back"] == 1 and s["redone"] == 1, s
    assert np.array_equal(ce.vmap_dump()["data"], co.vmap_dump()["data"]) and np.array_equal(ce.read_image(), co.read_image())
    assert ce.resize(290, 120) == L.LQR_OK and co.resize(290, 120) == L.LQR_OK        # back up inside the cached map
    assert np.array_equal(ce.read_image(), co.read_image())
    ce.destroy(); co.destroy()


def test_after_a_time_out_the_process_stays_off_the_spin_kernels(oracle, engine, lib):
    """a device that is shared or partitioned now will be in a minute: the time-out is paid once, later resizes go straight to the
    non-spinning kernels (and are exact); lqrhip_set_no_spin(0) re-arms the persistent ones"""
    img = D.photo_like(300, 160, 84)
    lib.lqrhip_debug_inject(1, 3, 1)
    assert lib.lqrhip_get_no_spin() == 0
    H.assert_same(H.run_case(oracle, img, 270, 150), H.run_case(engine, img, 270, 150), "time-out, recovered")
    assert lib.lqrhip_get_no_spin() == 1
    lib.lqrhip_prof_reset(); lib.lqrhip_prof_enable(1)
    H.assert_same(H.run_case(oracle, img, 250, 150), H.run_case(engine, img, 250, 150), "after the time-out")
    lib.lqrhip_prof_enable(0)
    ms, n, by = ctypes.c_double(0), ctypes.c_longlong(0), ctypes.c_double(0)
    lib.lqrhip_prof_get(b"dp_update_tiled", ctypes.byref(ms), ctypes.byref(n), ctypes.byref(by))
    assert n.value == 0, "a persistent kernel ran after the time-out"
    lib.lqrhip_set_no_spin(0)


def test_the_self_checks_can_be_switched_off(oracle, engine, lib):
    """lqrhip_set_selfcheck(0): the damaged seam log goes unnoticed (LQR_OK, a map that is not the genuine one) -- which is what
    the checks are there to prevent; it also shows that the injection really damages something"""
    img = D.photo_like(300, 160, 85)
    ref = H.run_case(oracle, img, 260, 160)
    lib.lqrhip_set_selfcheck(0)
    lib.lqrhip_debug_inject(3, 7, 1)
    got = H.run_case(engine, img, 260, 160)
    lib.lqrhip_set_selfcheck(1)
    assert got["ret"] == L.LQR_OK and not np.array_equal(got["vmap"]["data"], ref["vmap"]["data"])


def test_a_failed_level_check_in_the_second_sub_batch_leaves_the_first_where_it_was(oracle, engine, lib):
    """34 images on two sub-batch streams: the level check fails in the SECOND sub-batch's inflate pass.  No sub-batch adopts its inflated
    layout before every one has passed (lqrhip_inflate stages, lqrhip_inflate_commit adopts): the whole group is rolled back and redone"""
    lib.lqrhip_sub_batches.argtypes = [ctypes.c_int]
    w, h, n = 520, 140, 34
    if lib.lqrhip_sub_batches(n) < 2:
        pytest.skip("this process has no hardware queues for sub-batch streams")
    imgs = [D.photo_like(w, h, 500 + i) if i % 2 else D.noise(w, h, 500 + i) for i in range(n)]
    cs = [L.Carver(engine, im).configure() for im in imgs]
    stats(lib, reset=True)
    lib.lqrhip_debug_inject(5, 1, 1)              # let the first sub-batch's commit pass
    assert L.resize_batch(engine, cs, w - 31, h) == L.LQR_OK
    s = stats(lib)
    assert s["injected"] == 1 and s["levels"] == 1 and s["rolled_back"] == 2 and s["redone"] == 2, s
    for c, im in zip(cs, imgs):
        ref = H.run_case(oracle, im, w - 31, h)
        assert np.array_equal(c.vmap_dump()["data"], ref["vmap"]["data"]) and np.array_equal(c.read_image(), ref["image"])
    for c in cs:
        c.destroy()


@pytest.mark.parametrize("case", ["shrink-both-directions", "shrink-3-images", "enlarge-in-steps", "masks-delta2", "deeper-session", "shrink-34-images"])
def test_an_allocation_failure_anywhere_in_a_resize_leaves_consistent_carvers(oracle, engine, lib, case):
    """lqrhip_debug_fail_alloc(n): the nth device allocation of the resize fails, once -- for every n until the resize gets through
    (working planes, seam log, exchange areas, backtrack maps, the staging of the inflate pass AFTER the session's levels are
    committed, flatten, transpose, the second direction ...).  Each time: LQR_NOMEM -- the one value the plug-in tests for
    (src/render.c:42-46) -- and every carver still serves a consistent image: one of the states the genuine sequence passes through
    (the original, the first direction's / the first enlargement step's result); asked again, the resize finishes exactly."""
    w, h, nw, nh, n_images, kw, n_states, pre = {
        "shrink-both-directions": (200, 120, 170, 100, 1, {}, 2, []),                       # the original, the first direction's result
        "shrink-3-images": (200, 120, 170, 100, 3, {}, 2, []),
        "enlarge-in-steps": (120, 90, 230, 90, 1, dict(enl_step=140.0), 2, []),             # the original, the first enlargement step's result
        "masks-delta2": (220, 130, 190, 130, 1, dict(pres=D.ellipse_mask(220, 130), disc=D.band_mask(220, 130, 30, 55), rigmask=D.top_half_mask(220, 130),
                                                     rigidity=6.0, delta_x=2), 1, []),
        # an interactive-style carver (src/render.c:465-574): 200 -> 185 first, in a call of its own; the sweep runs over 185 -> 160, a deeper
        # session of the multi-size image -- its levels are committed before the inflate pass is staged, and a stale level would be taken for a
        # carved pixel by the next lay-out of the working planes (lqrhip_session_rollback after LQR_NOMEM as well)
        "deeper-session": (200, 120, 160, 120, 1, {}, 1, [(185, 120)]),
        # two sub-batch streams where the process has the queues: flatten, transpose and inflate stage every sub-batch before any adopts its
        # new planes (lqrhip_planes_commit), so a failure in the second leaves the first where it was; every 11th allocation point
        "shrink-34-images": (120, 80, 104, 70, 34, {}, 2, []),
    }[case]
    stride = 11 if n_images > 8 else 1
    imgs = [D.photo_like(w, h, 91 + i) for i in range(n_images)]
    states, final = [{} for im in imgs], []
    start = pre[-1] if pre else (w, h)

    def oracle_at(i, size):
        co, _ = H.init_carver(oracle, imgs[i], nw, nh, **kw)
        for ps in pre:
            assert co.resize(*ps) == L.LQR_OK
        if size != start:
            assert co.resize(*size) == L.LQR_OK
        return co
    for i in range(n_images):
        co = oracle_at(i, (nw, nh))
        final.append((co.read_image(), co.vmap_dump()["data"]))
        co.destroy()

    def state(i, size):     # what the genuine sequence holds at that size (one call from the start: the same sessions)
        if size not in states[i]:
            co = oracle_at(i, size)
            states[i][size] = co.read_image()
            co.destroy()
        return states[i][size]
    resize = lambda cs, size=(nw, nh): cs[0].resize(*size) if n_images == 1 else L.resize_batch(engine, cs, *size)
    failures, seen = 0, set()
    for n in range(0, 6000, stride):
        cs = [H.init_carver(engine, im, nw, nh, **kw)[0] for im in imgs]
        for ps in pre:
            assert resize(cs, ps) == L.LQR_OK
        lib.lqrhip_debug_fail_alloc(n)
        ret = resize(cs)
        lib.lqrhip_debug_fail_alloc(-1)
        if ret == L.LQR_OK:
            for c, (fi, fv) in zip(cs, final):
                assert np.array_equal(c.read_image(), fi) and np.array_equal(c.vmap_dump()["data"], fv), n
                c.destroy()
            break
        assert ret == L.LQR_NOMEM, (n, ret)
        failures += 1
        sizes = set()
        for i, c in enumerate(cs):
            g = c.getters()
            sizes.add((g["width"], g["height"]))
            assert np.array_equal(c.read_image(), state(i, (g["width"], g["height"]))), (n, g)
        assert len(sizes) == 1, (n, sizes)          # a lock-step group fails as one
        seen |= sizes
        assert resize(cs) == L.LQR_OK, n
        for c, (fi, fv) in zip(cs, final):
            assert np.array_equal(c.read_image(), fi) and np.array_equal(c.vmap_dump()["data"], fv), n
            c.destroy()
    else:
        pytest.fail("the resize never got through")
    print("allocation sweep %s: %d failure points, states seen %s" % (case, failures, sorted(seen)))
    assert failures >= 3 and len(seen) == n_states, (failures, seen)      # the sweep walked through the allocations of every stage
