"""ctypes binding of the LqrCarver C ABI (include/lqr.h) -- the Python-side mirror of the
interface gimp-lqr-plugin's src/render.c and src/io_functions.c consume.

`engine_api()` binds the MI355X engine (liblqr-hip.so, next to this file).  `Api(path, prefix)`
binds any other library exporting the same ABI (the tests bind their CPU oracle and, where one
exists, a genuine liblqr-1 this way -- nothing in this package knows about either).

Call order in `Carver.configure` mirrors the reference's render_init_carver
(src/render.c:220-248); read-out mirrors write_carver_to_layer (src/io_functions.c:155-164).
"""
import ctypes as C
import os

import numpy as np

try:        # if torch is going to be used in this process, its bundled HIP runtime must load first
    import torch  # noqa: F401
except Exception:      # the engine does not need torch
    torch = None

HERE = os.path.dirname(os.path.abspath(__file__))
# (LQR_HIP_LIB: another build of the same library, for A/B runs of bench.py and the tests)
ENGINE_LIB = os.environ.get("LQR_HIP_LIB") or os.path.join(HERE, "liblqr-hip.so")

LQR_ERROR, LQR_OK, LQR_NOMEM, LQR_USRCANCEL = 0, 1, 2, 3
LQR_RES_ORDER_HOR, LQR_RES_ORDER_VERT = 0, 1
(LQR_EF_GRAD_NORM, LQR_EF_GRAD_SUMABS, LQR_EF_GRAD_XABS, LQR_EF_LUMA_GRAD_NORM,
 LQR_EF_LUMA_GRAD_SUMABS, LQR_EF_LUMA_GRAD_XABS, LQR_EF_NULL) = range(7)

_libc = C.CDLL(None)
_libc.malloc.restype = C.c_void_p
_libc.malloc.argtypes = [C.c_size_t]

PROGRESS_INIT = C.CFUNCTYPE(C.c_int, C.c_char_p)
PROGRESS_UPDATE = C.CFUNCTYPE(C.c_int, C.c_double)
PROGRESS_END = C.CFUNCTYPE(C.c_int, C.c_char_p)
VMAP_FUNC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)

# name -> (restype, argtypes); every symbol include/lqr.h declares
_P, _I, _F, _D = C.c_void_p, C.c_int, C.c_float, C.c_double
SYMBOLS = {
    "lqr_carver_new": (_P, [_P, _I, _I, _I]),
    "lqr_carver_init": (_I, [_P, _I, _F]),
    "lqr_carver_destroy": (None, [_P]),
    "lqr_carver_attach": (_I, [_P, _P]),
    "lqr_carver_set_energy_function_builtin": (_I, [_P, _I]),
    "lqr_carver_set_resize_order": (None, [_P, _I]),
    "lqr_carver_set_progress": (None, [_P, _P]),
    "lqr_carver_set_side_switch_frequency": (None, [_P, C.c_uint]),
    "lqr_carver_set_enl_step": (_I, [_P, _F]),
    "lqr_carver_get_enl_step": (_F, [_P]),
    "lqr_carver_set_dump_vmaps": (None, [_P]),
    "lqr_carver_bias_add_rgb_area": (_I, [_P, _P, _I, _I, _I, _I, _I, _I]),
    "lqr_carver_rigmask_add_rgb_area": (_I, [_P, _P, _I, _I, _I, _I, _I]),
    "lqr_carver_resize": (_I, [_P, _I, _I]),
    "lqr_carver_flatten": (_I, [_P]),
    "lqr_carver_scan_line": (_I, [_P, C.POINTER(_I), C.POINTER(_P)]),
    "lqr_carver_scan_by_row": (_I, [_P]),
    "lqr_carver_scan_reset": (None, [_P]),
    "lqr_carver_get_width": (_I, [_P]),
    "lqr_carver_get_height": (_I, [_P]),
    "lqr_carver_get_channels": (_I, [_P]),
    "lqr_carver_get_ref_width": (_I, [_P]),
    "lqr_carver_get_ref_height": (_I, [_P]),
    "lqr_carver_get_orientation": (_I, [_P]),
    "lqr_carver_get_depth": (_I, [_P]),
    "lqr_carver_list_start": (_P, [_P]),
    "lqr_carver_list_current": (_P, [_P]),
    "lqr_carver_list_next": (_P, [_P]),
    "lqr_vmap_dump": (_P, [_P]),
    "lqr_vmap_destroy": (None, [_P]),
    "lqr_vmap_get_data": (C.POINTER(_I), [_P]),
    "lqr_vmap_get_width": (_I, [_P]),
    "lqr_vmap_get_height": (_I, [_P]),
    "lqr_vmap_get_depth": (_I, [_P]),
    "lqr_vmap_get_orientation": (_I, [_P]),
    "lqr_vmap_list_start": (_P, [_P]),
    "lqr_vmap_list_current": (_P, [_P]),
    "lqr_vmap_list_next": (_P, [_P]),
    "lqr_vmap_list_foreach": (_I, [_P, VMAP_FUNC, _P]),
    "lqr_progress_new": (_P, []),
    "lqr_progress_set_init": (_I, [_P, PROGRESS_INIT]),
    "lqr_progress_set_update": (_I, [_P, PROGRESS_UPDATE]),
    "lqr_progress_set_end": (_I, [_P, PROGRESS_END]),
    "lqr_progress_set_update_step": (_I, [_P, _F]),
    "lqr_progress_set_init_width_message": (_I, [_P, C.c_char_p]),
    "lqr_progress_set_init_height_message": (_I, [_P, C.c_char_p]),
    "lqr_progress_set_end_width_message": (_I, [_P, C.c_char_p]),
    "lqr_progress_set_end_height_message": (_I, [_P, C.c_char_p]),
    "lqrx_carver_get_energy": (_I, [_P, _P]),
    "lqrx_carver_frame_width": (_I, [_P]),
    "lqrx_carver_frame_height": (_I, [_P]),
    "lqrx_carver_debug_maps": (_I, [_P, _P, _P, _P]),
    "lqrx_set_debug": (None, [_I]),
    "lqrx_carver_debug_width": (_I, [_P]),
    "lqrx_carver_debug_height": (_I, [_P]),
    "lqrx_carver_debug_snapshot": (_I, [_P, _P, _P, _P]),
    "lqrx_carver_read_image": (_I, [_P, _P]),
    "lqrx_carver_resize_batch": (_I, [C.POINTER(_P), _I, _I, _I]),
    "lqrx_carver_read_image_device": (_I, [_P, _P]),
    "lqrx_guess_new_size": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I]),
    "lqrx_vmap_to_rgba": (_I, [_P, C.POINTER(_D), C.POINTER(_D), _P]),
    "lqrx_carver_reload_device_batch": (_I, [C.POINTER(_P), _I, C.POINTER(_P)]),
}
# liblqr-1 proper exports only the lqr_* part
LIBLQR_SYMBOLS = [s for s in SYMBOLS if s.startswith("lqr_")]


class Api:
    """Resolved function table of one library exporting the ABI."""

    def __init__(self, path, prefix="", symbols=None):
        self.path, self.prefix = path, prefix
        self.lib = C.CDLL(path, mode=C.RTLD_LOCAL)
        self.has_ext = True
        for name in (symbols or SYMBOLS):
            res, args = SYMBOLS[name]
            fn = getattr(self.lib, prefix + name)   # AttributeError = missing export
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)


_apis = {}


def engine_api():
    if "engine" not in _apis:
        if not os.path.exists(ENGINE_LIB):
            raise RuntimeError("HIP engine library missing: %s (run python -c 'import __graft_entry__ as g; g.build()')" % ENGINE_LIB)
        _apis["engine"] = Api(ENGINE_LIB, "")
    return _apis["engine"]


def _malloc_copy(arr):
    """The carver takes ownership of a malloc'd buffer (render.c:220-223)."""
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    p = _libc.malloc(max(arr.nbytes, 1))
    if not p:
        raise MemoryError
    C.memmove(p, arr.ctypes.data, arr.nbytes)
    return p


class Carver:
    """One LqrCarver driven through the C ABI."""

    def __init__(self, api, img, init=True, delta_x=1, rigidity=0.0):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        if img.ndim == 2:
            img = img[:, :, None]
        self.api = api
        self.h0, self.w0, self.ch = img.shape
        self._cbs = []
        self.events = []
        self.p = api.lqr_carver_new(_malloc_copy(img), self.w0, self.h0, self.ch)
        if not self.p:
            raise MemoryError("lqr_carver_new returned NULL")
        self.aux = []
        if init:
            ret = api.lqr_carver_init(self.p, delta_x, float(rigidity))
            assert ret == LQR_OK, ret

    @classmethod
    def from_buffer(cls, api, buf, w, h, ch, delta_x=1, rigidity=0.0):
        """lqr_carver_new + lqr_carver_init (render.c:222-224) on a malloc'ed interleaved u8 buffer the library takes
        ownership of (_malloc_copy): the two C calls and nothing else, for callers that time the upload"""
        self = cls.__new__(cls)
        self.api = api
        self.h0, self.w0, self.ch = h, w, ch
        self._cbs = []
        self.events = []
        self.aux = []
        self.p = api.lqr_carver_new(buf, w, h, ch)
        if not self.p:
            raise MemoryError("lqr_carver_new returned NULL")
        ret = api.lqr_carver_init(self.p, delta_x, float(rigidity))
        assert ret == LQR_OK, ret
        return self

    # -- configuration, in the order of render.c:225-248 ------------------
    def bias_add(self, mask, factor, x_off=0, y_off=0):
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        if mask.ndim == 2:
            mask = mask[:, :, None]
        h, w, ch = mask.shape
        return self.api.lqr_carver_bias_add_rgb_area(self.p, mask.ctypes.data, int(factor), ch, w, h, x_off, y_off)

    def rigmask_add(self, mask, x_off=0, y_off=0):
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        if mask.ndim == 2:
            mask = mask[:, :, None]
        h, w, ch = mask.shape
        return self.api.lqr_carver_rigmask_add_rgb_area(self.p, mask.ctypes.data, ch, w, h, x_off, y_off)

    def configure(self, nrg_func=LQR_EF_GRAD_XABS, res_order=LQR_RES_ORDER_HOR, switch_freq=2,
                  enl_step=1.5, dump_vmaps=False, progress=False):
        a = self.api
        assert a.lqr_carver_set_energy_function_builtin(self.p, nrg_func) == LQR_OK
        a.lqr_carver_set_resize_order(self.p, res_order)
        if progress:
            self.set_progress_recorder()
        a.lqr_carver_set_side_switch_frequency(self.p, switch_freq)
        assert a.lqr_carver_set_enl_step(self.p, float(enl_step)) == LQR_OK
        if dump_vmaps:
            a.lqr_carver_set_dump_vmaps(self.p)
        return self

    def set_progress_recorder(self):
        a = self.api
        prog = a.lqr_progress_new()
        ev = self.events
        cb_i = PROGRESS_INIT(lambda m: (ev.append(("init", m.decode())), 1)[1])
        cb_u = PROGRESS_UPDATE(lambda f: (ev.append(("update", f)), 1)[1])
        cb_e = PROGRESS_END(lambda m: (ev.append(("end", m.decode() if m else "")), 1)[1])
        self._cbs += [cb_i, cb_u, cb_e]
        a.lqr_progress_set_init(prog, cb_i)
        a.lqr_progress_set_update(prog, cb_u)
        a.lqr_progress_set_end(prog, cb_e)
        a.lqr_progress_set_init_width_message(prog, b"Resizing width...")
        a.lqr_progress_set_init_height_message(prog, b"Resizing height...")
        a.lqr_carver_set_progress(self.p, prog)

    def attach(self, img):
        """attach_aux_carver, render.c:881-900"""
        aux = Carver(self.api, img, init=False)
        ret = self.api.lqr_carver_attach(self.p, aux.p)
        assert ret == LQR_OK, ret
        self.aux.append(aux)
        return aux

    # -- run ---------------------------------------------------------------
    def resize(self, w1, h1):
        return self.api.lqr_carver_resize(self.p, int(w1), int(h1))

    def flatten(self):
        return self.api.lqr_carver_flatten(self.p)

    # -- readout: the loop of io_functions.c:155-164 -------------------------
    def read_scanlines(self):
        a = self.api
        W, H, ch = a.lqr_carver_get_width(self.p), a.lqr_carver_get_height(self.p), self.ch
        out = np.zeros((H, W, ch), np.uint8)
        n, line = C.c_int(0), C.c_void_p()
        count = 0
        a.lqr_carver_scan_reset(self.p)
        while a.lqr_carver_scan_line(self.p, C.byref(n), C.byref(line)):
            by_row = a.lqr_carver_scan_by_row(self.p)
            length = W if by_row else H
            buf = np.ctypeslib.as_array(C.cast(line, C.POINTER(C.c_ubyte)), shape=(length * ch,)).reshape(length, ch)
            if by_row:
                out[n.value] = buf
            else:
                out[:, n.value] = buf
            count += 1
        return out, count

    def read_image(self):
        a = self.api
        if not a.has_ext:
            return self.read_scanlines()[0]
        W, H = a.lqr_carver_get_width(self.p), a.lqr_carver_get_height(self.p)
        out = np.zeros((H, W, self.ch), np.uint8)
        assert a.lqrx_carver_read_image(self.p, out.ctypes.data) == LQR_OK
        return out

    def getters(self):
        a = self.api
        return dict(width=a.lqr_carver_get_width(self.p), height=a.lqr_carver_get_height(self.p),
                    channels=a.lqr_carver_get_channels(self.p), ref_width=a.lqr_carver_get_ref_width(self.p),
                    ref_height=a.lqr_carver_get_ref_height(self.p), orientation=a.lqr_carver_get_orientation(self.p),
                    depth=a.lqr_carver_get_depth(self.p), enl_step=a.lqr_carver_get_enl_step(self.p))

    def _vmap_to_dict(self, v):
        a = self.api
        w, h = a.lqr_vmap_get_width(v), a.lqr_vmap_get_height(v)
        data = np.ctypeslib.as_array(a.lqr_vmap_get_data(v), shape=(h * w,)).reshape(h, w).copy()
        return dict(data=data, depth=a.lqr_vmap_get_depth(v), orientation=a.lqr_vmap_get_orientation(v))

    def vmap_dump(self):
        v = self.api.lqr_vmap_dump(self.p)
        assert v
        d = self._vmap_to_dict(v)
        self.api.lqr_vmap_destroy(v)
        return d

    def dumped_vmaps(self):
        """write_all_vmaps, io_functions.c:292-314: foreach over the carver's list"""
        out = []

        def cb(v, _):
            out.append(self._vmap_to_dict(v))
            return LQR_OK
        fn = VMAP_FUNC(cb)
        ret = self.api.lqr_vmap_list_foreach(self.api.lqr_vmap_list_start(self.p), fn, None)
        assert ret == LQR_OK
        return out

    def energy(self):
        a = self.api
        ret_w = a.lqrx_carver_frame_width(self.p)
        ret_h = a.lqrx_carver_frame_height(self.p)
        buf = np.zeros((ret_h, ret_w), np.float32)
        ret = a.lqrx_carver_get_energy(self.p, buf.ctypes.data)
        assert ret == LQR_OK, ret
        w, h = a.lqrx_carver_frame_width(self.p), a.lqrx_carver_frame_height(self.p)
        if (w, h) != (ret_w, ret_h):      # the call flattened the carver
            buf = np.zeros((h, w), np.float32)
            assert a.lqrx_carver_get_energy(self.p, buf.ctypes.data) == LQR_OK
        return buf

    def debug_snapshot(self):
        a = self.api
        w, h = a.lqrx_carver_debug_width(self.p), a.lqrx_carver_debug_height(self.p)
        en, m, dx = np.zeros((h, w), np.float32), np.zeros((h, w), np.float32), np.zeros((h, w), np.int32)
        assert a.lqrx_carver_debug_snapshot(self.p, en.ctypes.data, m.ctypes.data, dx.ctypes.data) == LQR_OK
        return en, m, dx

    def destroy(self):
        if self.p:
            self.api.lqr_carver_destroy(self.p)   # also frees attached carvers
            self.p = None
            for x in self.aux:
                x.p = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def vmap_to_rgba(api, vmap_dict_or_ptr, col_start, col_end, carver=None):
    """write_vmap_to_layer's colour ramp (io_functions.c:249-279) over a dumped map; returns h x w x 4 u8.
    `vmap_dict_or_ptr` is a LqrVMap* (c_void_p / int)."""
    v = vmap_dict_or_ptr
    w, h = api.lqr_vmap_get_width(v), api.lqr_vmap_get_height(v)
    out = np.zeros((h, w, 4), np.uint8)
    cs = (C.c_double * 3)(*[float(x) for x in col_start])
    ce = (C.c_double * 3)(*[float(x) for x in col_end])
    ret = api.lqrx_vmap_to_rgba(v, cs, ce, out.ctypes.data)
    assert ret == LQR_OK, ret
    return out


def reload_device_batch(api, carvers, device_ptrs):
    arr = (C.c_void_p * len(carvers))(*[c.p for c in carvers])
    ptrs = (C.c_void_p * len(carvers))(*[int(p) for p in device_ptrs])
    return api.lqrx_carver_reload_device_batch(arr, len(carvers), ptrs)


def resize_batch(api, carvers, w1, h1):
    arr = (C.c_void_p * len(carvers))(*[c.p for c in carvers])
    return api.lqrx_carver_resize_batch(arr, len(carvers), int(w1), int(h1))
