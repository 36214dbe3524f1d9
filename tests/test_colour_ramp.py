"""The seam-map colour ramp: the oracle's lqrx_vmap_to_rgba restates the per-pixel loop of
write_vmap_to_layer, reference src/io_functions.c:249-279 -- arithmetic that IS in the reference tree, so this part
of the oracle is pinned to reference source, not to recollection.  Checked here against hand-computed known answers
and against an independent numpy (float64) restatement of the same five expressions."""
import ctypes as C

import numpy as np

import datasets as D
import harness as H
import lqr_ctypes as L


def ramp_numpy(vm, depth, cs, ce):
    """io_functions.c:263-271 in numpy float64 (IEEE double, one rounding per operation, like the C code built
    without FMA contraction); (guchar)(255 * x) truncates"""
    vm = vm.astype(np.int64)
    value = (depth + 1 - vm).astype(np.float64) / np.float64(depth + 1)
    out = np.zeros(vm.shape + (4,), np.uint8)
    for k in range(3):
        out[..., k] = (np.float64(255) * (value * np.float64(cs[k]) + (np.float64(1) - value) * np.float64(ce[k]))).astype(np.int64).astype(np.uint8)
    out[..., 3] = (np.float64(255) * (np.float64(0.5) * (np.float64(1) + value))).astype(np.int64).astype(np.uint8)
    out[vm == 0] = 0
    return out


def dumped(api, img, nw, nh, **kw):
    """(LqrVMap*, carver) pairs stay valid until the carver is destroyed: convert inside the callback"""
    c, _ = H.init_carver(api, img, nw, nh, output_seams=True, **kw)
    assert c.resize(nw, nh) == L.LQR_OK
    return c


def test_known_answers(oracle):
    # a 1 x 4 map dumped by hand: depth 3 -> value = (4 - vs) / 4
    c = L.Carver(oracle, D.noise(8, 1, 1)).configure(dump_vmaps=True)
    assert c.resize(5, 1) == L.LQR_OK                      # 3 seams on one row: levels 1, 2, 3 each once
    res = []

    def cb(v, _):
        res.append((L.vmap_to_rgba(oracle, v, (1.0, 0.0, 0.0), (0.0, 0.0, 1.0)), c._vmap_to_dict(v)))
        return L.LQR_OK
    fn = L.VMAP_FUNC(cb)
    assert oracle.lqr_vmap_list_foreach(oracle.lqr_vmap_list_start(c.p), fn, None) == L.LQR_OK
    rgba, vm = res[0]
    assert vm["depth"] == 3 and sorted(vm["data"].ravel().tolist()) == [0, 0, 0, 0, 0, 1, 2, 3]
    for x in range(8):
        vs = int(vm["data"][0, x])
        if vs == 0:
            assert rgba[0, x].tolist() == [0, 0, 0, 0]                      # :253-259 transparent
        else:
            # value 3/4, 2/4, 1/4 -> R = 255*value truncated, B = 255*(1-value), A = 255*0.5*(1+value)
            want = {1: [191, 0, 63, 223], 2: [127, 0, 127, 191], 3: [63, 0, 191, 159]}[vs]
            assert rgba[0, x].tolist() == want, (vs, rgba[0, x].tolist())
    c.destroy()


def test_oracle_matches_numpy_restatement(oracle):
    img = D.photo_like(200, 120, 9)
    c = dumped(oracle, img, 150, 90)
    seen = []

    def cb(v, _):
        d = c._vmap_to_dict(v)
        for cs, ce in [((1.0, 1.0, 0.0), (0.2, 0.0, 0.0)), ((0.3, 0.77, 0.123456789), (1.0, 0.5, 1e-3))]:
            got = L.vmap_to_rgba(oracle, v, cs, ce)
            assert np.array_equal(got, ramp_numpy(d["data"], d["depth"], cs, ce))
        seen.append(d["depth"])
        return L.LQR_OK
    fn = L.VMAP_FUNC(cb)
    assert oracle.lqr_vmap_list_foreach(oracle.lqr_vmap_list_start(c.p), fn, None) == L.LQR_OK
    assert seen == [50, 30]
    c.destroy()
