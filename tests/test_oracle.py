"""CPU tests of the oracle (oracle/lqr_oracle.c): known answers small enough to
check by hand or by an independent pure-Python restatement, and properties the
domain offers.  The reference ships no tests for this path;
parity is UNPINNED: these tests pin the oracle to its specification, not to liblqr.
"""
import numpy as np
import pytest

import datasets as D
import harness as H
import lqr_ctypes as L


# ---------------------------------------------------------------------------
# an independent, loop-level restatement of the first seam (small cases only)
# ---------------------------------------------------------------------------
def py_brightness(img):
    h, w, ch = img.shape
    b = np.zeros((h, w), np.float64)
    for y in range(h):
        for x in range(w):
            p = img[y, x]
            if ch <= 2:
                v = float(p[0]) / 255
            else:
                v = (float(p[0]) / 255 + float(p[1]) / 255 + float(p[2]) / 255) / 3
            if ch in (2, 4):
                v *= float(p[ch - 1]) / 255
            b[y, x] = v
    return b


def py_energy_xabs(img):
    b = py_brightness(img)
    h, w = b.shape
    e = np.zeros((h, w), np.float32)
    for y in range(h):
        for x in range(w):
            if w == 1:
                gx = 0.0
            elif x == 0:
                gx = b[y, 1] - b[y, 0]
            elif x < w - 1:
                gx = (b[y, x + 1] - b[y, x - 1]) / 2
            else:
                gx = b[y, x] - b[y, x - 1]
            e[y, x] = np.float32(abs(gx))
    return e


def py_first_seam(e, delta=1, leftright=0):
    h, w = e.shape
    m = np.zeros((h, w), np.float32)
    least = np.zeros((h, w), np.int32)
    m[0] = e[0]
    for y in range(1, h):
        for x in range(w):
            lo, hi = max(-x, -delta), min(w - 1 - x, delta)
            best, bdx = m[y - 1, x + lo], lo
            for dx in range(lo + 1, hi + 1):
                c = m[y - 1, x + dx]
                if c < best or (c == best and leftright == 1):
                    best, bdx = c, dx
            m[y, x] = np.float32(e[y, x] + best)
            least[y, x] = bdx
    best, bx = np.float32(2 ** 29), 0
    for x in range(w):
        if m[h - 1, x] < best or (m[h - 1, x] == best and leftright == 1):
            best, bx = m[h - 1, x], x
    seam = [0] * h
    x = bx
    for y in range(h - 1, -1, -1):
        seam[y] = x
        x += least[y, x]
    return seam, m


@pytest.mark.parametrize("seed,ch", [(1, 1), (2, 3), (3, 4), (4, 2)])
def test_first_seam_matches_python_restatement(oracle, seed, ch):
    img = D.noise(9, 7, seed, channels=ch)
    c = L.Carver(oracle, img).configure(switch_freq=0)
    e = c.energy()
    assert np.array_equal(e, py_energy_xabs(img))           # E4, bit-exact
    seam, _ = py_first_seam(e)
    assert c.resize(8, 7) == L.LQR_OK
    vm = c.vmap_dump()["data"]
    assert vm.shape == (7, 9)
    for y in range(7):
        assert list(np.nonzero(vm[y])[0]) == [seam[y]]
    c.destroy()


def test_energy_hand_checked_row(oracle):
    """1 x 4 grey image [0, 51, 255, 102]: b = [0, .2, 1, .4];
    xabs = [.2, .5, .1, .6] (one-sided at both ends, central /2 inside)"""
    img = np.array([[0, 51, 255, 102]], np.uint8)[:, :, None]
    c = L.Carver(oracle, img)
    e = c.energy()
    expect = np.array([[0.2, 0.5, 0.1, 0.6]])
    assert np.allclose(e, expect, atol=1e-7)
    c.destroy()


def test_flat_image_tie_rule_left_then_right(oracle):
    """all energies equal: leftright=0 picks the leftmost seam (x=0 on every row);
    with side switches the later seams hug the right edge"""
    img = np.full((6, 12, 3), 77, np.uint8)
    c = L.Carver(oracle, img).configure(switch_freq=0)
    assert c.resize(9, 6) == L.LQR_OK
    vm = c.vmap_dump()["data"]
    for y in range(6):      # three seams, each time the leftmost remaining column
        assert list(vm[y, :3]) == [1, 2, 3] and not vm[y, 3:].any()
    c.destroy()
    c = L.Carver(oracle, img).configure(switch_freq=2)       # interval = (4-1-1)//2+1 = 2: switch after seams 2, 4
    assert c.resize(8, 6) == L.LQR_OK
    vm = c.vmap_dump()["data"]
    for y in range(6):
        assert vm[y, 0] == 1            # seam 1: leftmost
        assert vm[y, 11] in (2, 3)      # after the first switch the rightmost column goes
    c.destroy()


def test_discard_mask_attracts_seams_null_energy(oracle):
    img = D.noise(20, 10, 5)
    disc = D.band_mask(20, 10, 6, 10)
    r = H.run_case(oracle, img, 16, 10, disc=disc, nrg_func=L.LQR_EF_NULL)
    vm = r["vmap"]["data"]
    assert (vm[:, 6:10] > 0).all() and not vm[:, :6].any() and not vm[:, 10:].any()


def test_preserve_mask_repels_seams(oracle):
    img = D.noise(24, 12, 6)
    pres = D.band_mask(24, 12, 8, 16)
    r = H.run_case(oracle, img, 16, 12, pres=pres)
    assert not r["vmap"]["data"][:, 8:16].any()


@pytest.mark.parametrize("delta,rigidity", [(1, 0.0), (2, 0.0), (2, 10.0), (3, 1.0), (0, 0.0)])
def test_seams_are_connected_and_remove_one_pixel_per_row(oracle, delta, rigidity):
    img = D.photo_like(40, 30, 11)
    n = 12
    c = L.Carver(oracle, img, delta_x=delta, rigidity=rigidity).configure()
    assert c.resize(40 - n, 30) == L.LQR_OK
    vm = c.vmap_dump()["data"]
    out = c.read_image()
    remaining = np.ones((30, 40), bool)
    for k in range(1, n + 1):
        xs = []
        for y in range(30):
            cols = np.nonzero(vm[y] == k)[0]
            assert len(cols) == 1
            # position in the frame as it was when seam k was carved
            xs.append(int(remaining[y, :cols[0]].sum()))
        for y in range(30):
            remaining[y, np.nonzero(vm[y] == k)[0][0]] = False
        assert all(abs(xs[y] - xs[y - 1]) <= delta for y in range(1, 30)), (k, xs)
    for y in range(30):
        assert np.array_equal(out[y], img[y][vm[y] == 0])
    c.destroy()


def test_shrink_then_restore_is_identity_and_cached(oracle):
    img = D.photo_like(50, 30, 12)
    c = L.Carver(oracle, img).configure(progress=True)
    assert c.resize(40, 30) == L.LQR_OK
    small = c.read_image()
    n_events = len(c.events)
    assert c.resize(50, 30) == L.LQR_OK and np.array_equal(c.read_image(), img)
    assert c.resize(45, 30) == L.LQR_OK     # inside the cached map: no new seams
    assert c.getters()["depth"] == 10
    assert c.resize(40, 30) == L.LQR_OK and np.array_equal(c.read_image(), small)
    assert len(c.events) > n_events
    c.destroy()


def test_enlarge_inserts_interpolated_seams(oracle):
    img = D.photo_like(30, 20, 13)
    c = L.Carver(oracle, img).configure()
    assert c.resize(36, 20) == L.LQR_OK
    big = c.read_image()
    assert big.shape == (20, 36, 4)
    vm = c.vmap_dump()["data"]
    for y in range(20):     # every original pixel survives, in order
        it = iter(range(36))
        for x in range(30):
            assert any(np.array_equal(big[y, j], img[y, x]) for j in it)
        assert (vm[y] > 0).sum() == 6
    g = c.getters()
    assert g["depth"] == 6 and g["ref_width"] == 30 and g["width"] == 36
    c.destroy()


def test_flatten_makes_current_size_the_reference(oracle):
    img = D.photo_like(40, 25, 14)
    c = L.Carver(oracle, img).configure()
    assert c.resize(32, 25) == L.LQR_OK
    small = c.read_image()
    assert c.flatten() == L.LQR_OK
    g = c.getters()
    assert (g["ref_width"], g["depth"], g["width"]) == (32, 0, 32)
    assert np.array_equal(c.read_image(), small)
    assert c.flatten() == L.LQR_OK and np.array_equal(c.read_image(), small)     # idempotent
    c.destroy()


def test_bidirectional_is_width_then_height(oracle):
    img = D.photo_like(40, 30, 15)
    r = H.run_case(oracle, img, 34, 24)
    assert r["image"].shape == (24, 34, 4)
    g = r["getters"]
    assert g["orientation"] == 1 and g["ref_width"] == 34 and g["ref_height"] == 30 and g["depth"] == 6
    # height-first gives a different decomposition
    r2 = H.run_case(oracle, img, 34, 24, res_order=L.LQR_RES_ORDER_VERT)
    assert r2["getters"]["orientation"] == 0 and r2["image"].shape == (24, 34, 4)


def test_scan_line_contract(oracle):
    """io_functions.c:155-164: every line exactly once, FALSE at the end, then rewound"""
    img = D.noise(12, 9, 16)
    c = L.Carver(oracle, img).configure()
    assert c.resize(10, 9) == L.LQR_OK
    a, n1 = c.read_scanlines()
    b, n2 = c.read_scanlines()
    assert n1 == n2 == 9 and np.array_equal(a, b)
    assert c.resize(10, 7) == L.LQR_OK
    a, n = c.read_scanlines()          # transposed carver: lines are image columns
    assert n == 10 and a.shape == (7, 10, 4) and oracle.lqr_carver_scan_by_row(c.p) == 0
    c.destroy()


def test_aux_carvers_follow_the_root(oracle):
    img = D.photo_like(36, 24, 17)
    pres = D.ellipse_mask(36, 24)
    r = H.run_case(oracle, img, 30, 20, pres=pres, resize_aux_layers=True, output_seams=True)
    assert len(r["aux"]) == 1 and r["aux"][0].shape == (20, 30, 4)
    assert len(r["vmaps"]) == 2         # one per direction (render.c:241,344)
    assert r["vmaps"][0]["orientation"] == 0 and r["vmaps"][1]["orientation"] == 1


def test_progress_events(oracle):
    img = D.noise(30, 20, 18)
    c = L.Carver(oracle, img).configure(progress=True)
    assert c.resize(20, 16) == L.LQR_OK
    kinds = [e[0] for e in c.events]
    assert kinds[0] == "init" and c.events[0][1] == "Resizing width..."
    assert kinds.count("init") == 2 and kinds.count("end") == 2
    ups = [e[1] for e in c.events if e[0] == "update"]
    assert all(0 <= u <= 1 for u in ups) and len(ups) == 10 + 4
    c.destroy()


def test_error_returns(oracle):
    img = D.noise(8, 6, 19)
    c = L.Carver(oracle, img)
    assert c.resize(0, 6) == L.LQR_ERROR and c.resize(8, -1) == L.LQR_ERROR
    assert oracle.lqr_carver_set_enl_step(c.p, 1.0) == L.LQR_ERROR
    assert oracle.lqr_carver_set_enl_step(c.p, 2.5) == L.LQR_ERROR
    assert oracle.lqr_carver_init(c.p, 1, 0.0) == L.LQR_ERROR       # already initialised
    aux = L.Carver(oracle, D.noise(7, 6, 1), init=False)
    assert oracle.lqr_carver_attach(c.p, aux.p) == L.LQR_ERROR      # size mismatch
    aux.destroy()
    c.destroy()


def test_resize_to_width_one(oracle):
    img = D.noise(6, 5, 20)
    c = L.Carver(oracle, img).configure()
    assert c.resize(1, 5) == L.LQR_OK
    assert c.read_image().shape == (5, 1, 4)
    assert c.resize(6, 5) == L.LQR_OK and np.array_equal(c.read_image(), img)
    c.destroy()


def test_enum_order_is_abi():
    assert (L.LQR_ERROR, L.LQR_OK, L.LQR_NOMEM) == (0, 1, 2)
    assert L.LQR_EF_GRAD_XABS == 2 and L.LQR_EF_LUMA_GRAD_NORM == 3 and L.LQR_EF_NULL == 6


def _py_guess(mask, x_off, y_off, ow, oh, direction):
    """loop-level restatement of guess_new_size (src/layers_combo.c:275-392)"""
    h, w, ch = mask.shape
    has_alpha = ch in (2, 4)
    cb = ch - (1 if has_alpha else 0)
    best = 0
    lines = range(max(0, y_off), min(oh, h + y_off)) if direction == 0 else range(max(0, x_off), min(ow, w + x_off))
    n = (min(ow, w + x_off) - max(0, x_off)) if direction == 0 else (min(oh, h + y_off) - max(0, y_off))
    for z1 in lines:
        cnt = 0
        for z2 in range(max(n, 0)):
            px = mask[z1 - y_off, max(0, -x_off) + z2] if direction == 0 else mask[max(0, -y_off) + z2, z1 - x_off]
            s = float(sum(int(v) for v in px[:cb])) / (255 * cb)
            if has_alpha:
                s *= float(px[ch - 1]) / 255
            cnt += s >= 0.5 / cb
        best = max(best, cnt)
    return (oh if direction else ow) - best


@pytest.mark.parametrize("ch,x_off,y_off", [(4, 0, 0), (4, -5, 3), (3, 7, -4), (2, 2, 2), (1, -3, -3)])
def test_guess_new_size_matches_restatement(oracle, ch, x_off, y_off):
    rng = np.random.default_rng(100 + ch + x_off)
    mask = rng.integers(0, 256, size=(23, 31, ch), dtype=np.uint8)
    mask[5:15, 8:20] = 255
    for direction in (0, 1):
        got = oracle.lqrx_guess_new_size(mask.ctypes.data, ch, 31, 23, x_off, y_off, 28, 20, direction)
        assert got == _py_guess(mask, x_off, y_off, 28, 20, direction)


def test_null_energy_with_masks_never_leaves_dangling_back_pointers(oracle):
    """found by scripts/fuzz_parity.py: with the null energy function and preserve/discard masks the maps are
    so heavily tied that update_mmap's band used to shrink past the children of the carved pixel, leaving a
    back pointer to a pixel that no longer exists (spec delta 6)"""
    import ctypes
    import datasets as D
    import harness as H
    w, h = 276, 80
    img = D.noise(w, h, 790234955, channels=1)
    oracle.lib.olqrx_set_debug.argtypes = [ctypes.c_int]
    oracle.lib.olqr_oracle_get_stats.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
    oracle.lib.olqrx_set_debug(1)
    oracle.lib.olqr_oracle_reset_stats()
    try:
        r = H.run_case(oracle, img, 260, 55, nrg_func=6, switch_freq=2, res_order=0,
                       pres=D.ellipse_mask(w, h), disc=D.band_mask(w, h, w // 5, w // 3))
    finally:
        oracle.lib.olqrx_set_debug(0)
    st = (ctypes.c_longlong * 8)()
    oracle.lib.olqr_oracle_get_stats(st)
    assert r["ret"] == 1 and r["image"].shape[:2] == (55, 260)
    assert st[7] == 0, "back pointers to carved pixels: %d" % st[7]


# ---------------------------------------------------------------------------
# an independent restatement of a whole session (several seams): energy is recomputed from
# scratch for every seam and update_mmap's keep-rule is applied to EVERY pixel instead of
# liblqr's band -- by DESIGN.md section 4.4 (any superset of the pixels with changed inputs
# gives the same memory) both must agree bit for bit, here and in the GPU kernels that rely
# on it (k_dp_tile_p<UPDATE>, k_band_update_tw)
# ---------------------------------------------------------------------------
def py_session(img, n_seams, leftright=0, switch_freq=0):
    """switch_freq: number of side switches per session (spec delta 1): at a switch the
    seam is still picked with the old tie rule, then the map is rebuilt from scratch with the new one"""
    h, w, ch = img.shape
    interval = (n_seams - 1) // switch_freq + 1 if switch_freq else 0
    rows = [[(y, x) for x in range(w)] for y in range(h)]            # surviving pixels, by id
    m, least = {}, {}

    def energy():
        cur = np.array([[img[p] for p in r] for r in rows], dtype=np.uint8).reshape(h, len(rows[0]), ch)
        return py_energy_xabs(cur)

    lr = [leftright]

    def best_parent(y, x):
        leftright = lr[0]
        wc = len(rows[y])
        cands = [xx for xx in (x - 1, x, x + 1) if 0 <= xx < wc]
        bx = cands[0]
        for xx in cands[1:]:
            a, b = m[rows[y - 1][xx]], m[rows[y - 1][bx]]
            if a < b or (a == b and leftright == 1):
                bx = xx
        return rows[y - 1][bx]

    e = energy()
    for x, p in enumerate(rows[0]):
        m[p] = e[0, x]
    for y in range(1, h):
        for x, p in enumerate(rows[y]):
            q = best_parent(y, x)
            least[p] = q
            m[p] = np.float32(e[y, x] + m[q])
    seams = []
    for s in range(n_seams):
        wc = len(rows[0])
        best, bx = np.float32(2 ** 29), 0
        for x in range(wc):
            v = m[rows[h - 1][x]]
            if v < best or (v == best and lr[0] == 1):
                best, bx = v, x
        rebuild = bool(interval) and (s + interval // 2) % interval == 0
        if rebuild:
            lr[0] ^= 1
        p = rows[h - 1][bx]
        seam = [None] * h
        for y in range(h - 1, -1, -1):
            seam[y] = p
            if y > 0:
                p = least[p]
                assert p in rows[y - 1], "dangling back pointer"
        seams.append(seam)
        for y in range(h):
            rows[y].remove(seam[y])
        if len(rows[0]) <= 1:
            break
        e = energy()
        for x, p in enumerate(rows[0]):
            m[p] = e[0, x]
        for y in range(1, h):
            for x, p in enumerate(rows[y]):
                q = best_parent(y, x)
                new_m = np.float32(e[y, x] + m[q])
                if not rebuild and least[p] == q and float(abs(np.float32(m[p] - new_m))) < 1e-5:
                    pass                                   # the stale value is kept
                else:
                    m[p] = new_m
                least[p] = q
    out = np.array([[img[p] for p in r] for r in rows], dtype=np.uint8).reshape(h, len(rows[0]), ch)
    return out, seams


@pytest.mark.parametrize("gen,seed,ch", [("noise", 5, 4), ("photo_like", 6, 3), ("flat_blocks", 7, 1), ("noise", 8, 2)])
def test_session_matches_python_restatement_with_full_width_keep_rule(oracle, gen, seed, ch):
    w, h, n = 26, 14, 9
    img = getattr(D, gen)(w, h, seed, channels=ch)
    out, seams = py_session(img, n)
    r = H.run_case(oracle, img, w - n, h, switch_freq=0)
    assert r["ret"] == L.LQR_OK
    assert np.array_equal(r["image"], out)
    # seam k of the session carries the k-th distinct non-zero level of the visibility map
    vm = r["vmap"]["data"]
    levels = sorted(set(int(v) for v in np.unique(vm) if v != 0))
    assert len(levels) == n
    for k, seam in enumerate(seams):
        ys, xs = np.nonzero(vm == levels[k])
        assert sorted(zip(ys.tolist(), xs.tolist())) == sorted(seam)


@pytest.mark.parametrize("gen,seed,ch", [("noise", 15, 3), ("photo_like", 16, 4)])
def test_vertical_session_is_the_transposed_horizontal_one(oracle, gen, seed, ch):
    """E11: a height change is the same session on the transposed image (src/render.c resizes width, then
    height; liblqr transposes the carver in between)"""
    w, h, n = 15, 22, 6
    img = getattr(D, gen)(w, h, seed, channels=ch)
    out_t, _ = py_session(np.ascontiguousarray(img.transpose(1, 0, 2)), n)
    r = H.run_case(oracle, img, w, h - n, switch_freq=0)
    assert r["ret"] == L.LQR_OK
    assert np.array_equal(r["image"], out_t.transpose(1, 0, 2))


@pytest.mark.parametrize("freq", [1, 2, 3, 100])
def test_side_switch_schedule_matches_python_restatement(oracle, freq):
    """the tie rule flips `freq` times per session, each flip with a full rebuild (freq >= seams: every seam)"""
    w, h, n = 24, 12, 8
    img = D.flat_blocks(w, h, 21, channels=3)          # many ties: the tie rule matters
    out, _ = py_session(img, n, switch_freq=freq)
    r = H.run_case(oracle, img, w - n, h, switch_freq=freq)
    assert np.array_equal(r["image"], out)
