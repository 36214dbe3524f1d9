"""The genuine engine driven one seam at a time: lqr_carver_build_maps / build_vsmap (lqr_carver.c:537-561,703-795 of
liblqr 0.4.1; genuine code at 0x414a80 / 0x4148d0) restated as a Python loop over the exe's own internal entry points, so
that the energy and DP planes can be read out of its memory at any point.  Build container only."""
import struct

import numpy as np

import ref_engine as R

# struct _LqrCarver (the exe's stabs): byte offsets
OFF = dict(w_start=0, h_start=4, w=8, h=12, w0=16, h0=20, level=24, max_level=28, transposed=52, active=56, root=64,
           rigidity=84, rigidity_map=88, rigidity_mask=92, delta_x=96, rgb=100, vs=104, en=108, bias=112, m=116, least=120,
           _raw=124, raw=128, vpath=140, vpath_x=144, leftright=148, lr_switch_frequency=152, enl_step=156,
           nrg_xmin=196, nrg_xmax=200, nrg_uptodate=204, rcache=208)


class Stepper:
    def __init__(self, carver):
        self.c, self.r, self.p = carver, carver.r, carver.p

    def get(self, name, fmt="<i"):
        return struct.unpack(fmt, self.r.read(self.p + OFF[name], 4))[0]

    def put(self, name, v):
        self.r.write(self.p + OFF[name], struct.pack("<i", v))

    def arr(self, name, dtype, n, index0=0):
        ptr = self.get(name, "<I")
        if not ptr:
            return None
        return np.frombuffer(self.r.read(ptr + 4 * index0, n * np.dtype(dtype).itemsize), dtype).copy()

    def begin(self, depth):
        """build_maps up to build_vsmap's loop; returns the seam levels the loop will run"""
        r = self.r
        max_level = self.get("max_level")
        assert depth > max_level and self.get("active") and not self.get("root")
        r.call("lqr_carver_set_width", self.p, self.get("w_start") - max_level + 1)
        assert r.call("lqr_carver_build_emap", self.p) == 1
        assert r.call("lqr_carver_build_mmap", self.p) == 1
        freq = self.get("lr_switch_frequency")
        self.interval = (depth - max_level - 1) // freq + 1 if freq else 0
        self.l, self.depth = max_level, depth
        return range(max_level, depth)

    def seam(self):
        """one turn of build_vsmap's loop (lqr_carver.c:747-776); returns vpath_x of the seam"""
        r, p = self.r, self.p
        max_level = self.get("max_level")
        l = self.l
        r.call("lqr_carver_build_vpath", p)
        vx = self.arr("vpath_x", np.int32, self.get("h"))
        r.call("lqr_carver_update_vsmap", p, l + max_level - 1)
        self.put("level", self.get("level") + 1)
        self.put("w", self.get("w") - 1)
        r.call("lqr_carver_carve", p)
        if self.get("w") > 1:
            assert r.call("lqr_carver_update_emap", p) == 1
            if self.interval and (l - max_level + self.interval // 2) % self.interval == 0:
                self.put("leftright", self.get("leftright") ^ 1)
                assert r.call("lqr_carver_build_mmap", p) == 1
            else:
                assert r.call("lqr_carver_update_mmap", p) == 1
        else:
            r.call("lqr_carver_finish_vsmap", p)
        self.l += 1
        return vx

    def planes(self):
        """en, m, back pointer as dx -- in the carved frame, like the oracle's lqrx_carver_debug_maps"""
        w, h, w0, h0 = (self.get(k) for k in ("w", "h", "w0", "h0"))
        n = w0 * h0
        raw_base = self.get("_raw", "<I")
        rows = self.arr("raw", np.uint32, h)
        flat = np.frombuffer(self.r.read(raw_base, 4 * n), np.int32)
        ids = np.stack([flat[(int(rp) - raw_base) // 4:(int(rp) - raw_base) // 4 + w] for rp in rows])
        en, m, least = self.arr("en", np.float32, n), self.arr("m", np.float32, n), self.arr("least", np.int32, n)
        dx = np.zeros((h, w), np.int32)
        delta = self.get("delta_x")
        for y in range(1, h):
            pos = np.full(n, -10 ** 6, np.int64)
            pos[ids[y - 1]] = np.arange(w)
            d = pos[least[ids[y]]] - np.arange(w)
            d[np.abs(d) > delta] = -999            # a back pointer to a pixel that is not within reach any more (stale)
            dx[y] = d
        return en[ids], m[ids], dx

    def table(self, name, n, index0=0):
        return self.arr(name, np.float32, n, index0)
