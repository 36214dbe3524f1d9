"""Driver of refrun: the reference author's own liblqr build, executed in a seccomp-strict child.

BUILD CONTAINER ONLY (needs /root/reference and gcc -m32).  See refrun.c.  Nothing is copied from the
exe into the repository; the zip is unpacked under a scratch directory outside it.

`RefApi` gives the engine's entry points by name (addresses from the exe's own COFF symbol table);
`RefCarver` mirrors gimp-lqr-plugin_amd/binding.py's `Carver` method for method, so tests/harness.py's
run_case() drives the genuine code through the same call sequence as the oracle and the HIP engine.
"""
import os
import struct
import subprocess
import sys
import zipfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ZIP = "/root/reference/windows_installer_files/lqr-pack4win/.zip"
EXE_IN_ZIP = "lib/gimp/2.0/plug-ins/gimp-lqr-plugin.exe"
SCRATCH = os.environ.get("REF_ENGINE_SCRATCH", "/tmp/ref_engine")

(OP_WRITE, OP_READ, OP_ALLOC, OP_FREE, OP_CALL, OP_INFO, OP_SETCW, OP_EVENTS, OP_HEAPCHECK, OP_SCANALL, OP_POISON,
 OP_PROGRET, OP_QUIT, OP_WRAP) = range(1, 15)

LQR_ERROR, LQR_OK, LQR_NOMEM, LQR_USRCANCEL = 0, 1, 2, 3


def available():
    return os.path.exists(REF_ZIP)


def exe_bytes():
    with zipfile.ZipFile(REF_ZIP) as z:
        return z.read(EXE_IN_ZIP)


def build_runner():
    os.makedirs(SCRATCH, exist_ok=True)
    out = os.path.join(SCRATCH, "refrun")
    src = os.path.join(HERE, "refrun.c")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-m32", "-O1", "-nostdlib", "-static", "-ffreestanding", "-fno-stack-protector",
                               "-fno-pic", "-no-pie", "-fno-builtin", "-Wall", "-Wl,--build-id=none", "-o", out, src])
    return out


class PE:
    """the little of PE32/COFF that is needed: sections, import slots, the symbol table"""

    def __init__(self, data):
        self.data = data
        pe = struct.unpack_from("<I", data, 0x3c)[0]
        assert data[pe:pe + 4] == b"PE\0\0"
        (machine, nsec, _, symoff, nsym, optsz, _) = struct.unpack_from("<HHIIIHH", data, pe + 4)
        assert machine == 0x14c
        opt = pe + 24
        assert struct.unpack_from("<H", data, opt)[0] == 0x10b
        self.image_base = struct.unpack_from("<I", data, opt + 28)[0]
        self.size_of_image = struct.unpack_from("<I", data, opt + 56)[0]
        ndir = struct.unpack_from("<I", data, opt + 92)[0]
        dirs = [struct.unpack_from("<II", data, opt + 96 + 8 * i) for i in range(ndir)]
        self.sections = []
        so = opt + optsz
        for i in range(nsec):
            name, vsz, va, rsz, roff = struct.unpack_from("<8sIIII", data, so + 40 * i)
            self.sections.append((name.rstrip(b"\0").decode(), va, vsz, roff, rsz))
        # imports
        self.imports = {}       # name -> IAT slot VA
        self.all_slots = []
        rva, _ = dirs[1]
        off = self.rva2off(rva)
        while True:
            ilt, _, _, name_rva, iat = struct.unpack_from("<IIIII", data, off)
            if not name_rva:
                break
            off += 20
            dll = self.cstr(self.rva2off(name_rva))
            thunk = self.rva2off(ilt or iat)
            k = 0
            while True:
                ent = struct.unpack_from("<I", data, thunk + 4 * k)[0]
                if not ent:
                    break
                slot = self.image_base + iat + 4 * k
                self.all_slots.append(slot)
                if not ent & 0x80000000:
                    self.imports[self.cstr(self.rva2off(ent) + 2)] = slot
                k += 1
        # COFF symbols
        self.symbols = {}
        strtab = symoff + 18 * nsym
        i = 0
        while i < nsym:
            raw, value, sec, typ, scl, naux = struct.unpack_from("<8sIhHBB", data, symoff + 18 * i)
            if raw[:4] == b"\0\0\0\0":
                name = self.cstr(strtab + struct.unpack_from("<I", raw, 4)[0])
            else:
                name = raw.rstrip(b"\0").decode("latin1")
            if sec > 0 and scl in (2, 3) and name.startswith("_"):
                self.symbols.setdefault(name[1:], self.image_base + self.sections[sec - 1][1] + value)
            i += 1 + naux

    def cstr(self, off):
        end = self.data.index(b"\0", off)
        return self.data[off:end].decode("latin1")

    def rva2off(self, rva):
        for _, va, vsz, roff, rsz in self.sections:
            if va <= rva < va + max(vsz, rsz):
                return roff + rva - va
        raise ValueError(hex(rva))


# float-only functions of the engine (no double arithmetic inside): run under a 24-bit control word in "sse" mode
FLOAT_ONLY = ("lqr_carver_build_mmap", "lqr_carver_update_mmap", "lqr_carver_init", "lqr_carver_transpose", "lqr_carver_inflate",
              # enl_step and progress-step arithmetic (float x int): 200 * 0.02f is 4 in float, 3.9999999 at 53 bits
              "lqr_carver_resize_width", "lqr_carver_resize_height")
# ... which call this one, whose callees compute energies in double: back to the 53-bit control word inside it
DOUBLE_INSIDE = ("lqr_carver_build_maps",)


class Runner:
    """cw: the x87 control word the engine's code runs under.
         0x37f  what the exe's own CRT leaves (fninit, 64-bit mantissa): the build as shipped
         0x27f  53-bit mantissa: double arithmetic as an SSE2 build performs it
       float24: additionally run the FLOAT_ONLY functions under 0x07f (24-bit mantissa), i.e. float arithmetic as an SSE2
       build performs it ("sse" mode = cw 0x27f + float24)"""

    def __init__(self, cw=0x37f, poison=None, float24=False, float24_only=None):
        """float24_only: run only THESE functions of FLOAT_ONLY under the 24-bit word (per-function evidence, precision_evidence.py)"""
        self.pe = PE(exe_bytes())
        self.proc = subprocess.Popen([build_runner()], stdin=subprocess.PIPE, stdout=subprocess.PIPE, bufsize=0)
        pe = self.pe
        for name, va, vsz, roff, rsz in pe.sections:
            if name in (".text", ".data", ".rdata", ".idata") and rsz:
                self.write(pe.image_base + va, pe.data[roff:roff + min(rsz, vsz) if vsz else rsz])
        info = self.info()
        stubs = dict(g_try_malloc=info[0], g_try_malloc0=info[1], g_free=info[2], g_strlcpy=info[3],
                     g_atomic_int_add=info[4], g_atomic_int_exchange_and_add=info[5], g_usleep=info[6],
                     pow=info[7], _assert=info[8])
        self.cb_init, self.cb_update, self.cb_end = info[10], info[11], info[12]
        self.c_pow = info[7]
        for slot in pe.all_slots:
            self.write(slot, struct.pack("<I", info[9]))           # anything else the exe imports: trap
        for name, addr in stubs.items():
            self.write(pe.imports[name], struct.pack("<I", addr))
        self.set_cw(cw)
        self.wrapped = {}
        if float24 or float24_only:
            for k, name in enumerate(FLOAT_ONLY):
                if float24_only is None or name in float24_only:
                    self.wrap(k, name, 0x07f)
            for k, name in enumerate(DOUBLE_INSIDE):
                self.wrap(len(FLOAT_ONLY) + k, name, cw)
        if poison is not None:
            self._cmd(OP_POISON, poison)
            self._read(4)

    # -- wire ---------------------------------------------------------------
    def _cmd(self, op, a=0, b=0, c=0, payload=b""):
        self.proc.stdin.write(struct.pack("<IIII", op, a & 0xffffffff, b & 0xffffffff, c & 0xffffffff) + payload)

    def _read(self, n):
        out = bytearray()
        while len(out) < n:
            chunk = self.proc.stdout.read(n - len(out))
            if not chunk:
                rc = self.proc.wait()
                raise RefCrash("reference engine process ended (status %s)" % rc, rc)
            out += chunk
        return bytes(out)

    def write(self, addr, data):
        data = bytes(data)
        self._cmd(OP_WRITE, addr, len(data), 0, data)
        self._read(4)

    def read(self, addr, n):
        if n == 0:
            return b""
        self._cmd(OP_READ, addr, n)
        return self._read(n)

    def alloc(self, n, zero=False):
        self._cmd(OP_ALLOC, n, int(zero))
        p = struct.unpack("<I", self._read(4))[0]
        if not p:
            raise MemoryError
        return p

    def free(self, p):
        self._cmd(OP_FREE, p)
        self._read(4)

    def info(self):
        self._cmd(OP_INFO)
        return struct.unpack("<16I", self._read(64))

    def set_cw(self, cw):
        self._cmd(OP_SETCW, cw)
        self._read(4)

    def wrap(self, slot, name, cw):
        """retarget every `call rel32` to `name` inside the exe's code at a stub that runs it under control word `cw`"""
        pe = self.pe
        target = pe.symbols[name]
        self._cmd(OP_WRAP, slot, target, cw)
        stub = struct.unpack("<I", self._read(4))[0]
        text = next(s for s in pe.sections if s[0] == ".text")
        base = pe.image_base + text[1]
        code = pe.data[text[3]:text[3] + text[2]]
        lo, hi = pe.symbols["lqr_carver_set_image_type"] - base, pe.symbols["lqr_grad_xabs"] + 0x40 - base   # the engine's code
        n = 0
        i = code.find(b"\xe8", lo)
        while 0 <= i < hi:
            rel = struct.unpack_from("<i", code, i + 1)[0]
            if base + i + 5 + rel == target:
                self.write(base + i + 1, struct.pack("<i", stub - (base + i + 5)))
                n += 1
            i = code.find(b"\xe8", i + 1)
        self.wrapped[name] = stub
        return n

    def set_progress_return(self, v):
        self._cmd(OP_PROGRET, v)
        self._read(4)

    def call(self, name_or_addr, *args, fp=False):
        """cdecl call.  ints/pointers -> one word; ('f', x) -> float word; ('d', x) -> two words"""
        if isinstance(name_or_addr, str):
            addr = self.wrapped.get(name_or_addr) or self.pe.symbols[name_or_addr]
        else:
            addr = name_or_addr
        words = b""
        for a in args:
            if isinstance(a, tuple):
                words += struct.pack("<f" if a[0] == "f" else "<d", a[1])
            else:
                words += struct.pack("<I", int(a) & 0xffffffff)
        self._cmd(OP_CALL, addr, len(words) // 4, int(fp), words)
        eax, st0 = struct.unpack("<Id", self._read(12))
        return st0 if fp else eax

    def calls(self, name, *args):
        v = self.call(name, *args)
        return v - (1 << 32) if v & 0x80000000 else v

    def events(self):
        self._cmd(OP_EVENTS)
        n = struct.unpack("<I", self._read(4))[0]
        raw = self._read(n * 64)
        out = []
        for i in range(n):
            kind, val, msg = struct.unpack_from("<Id52s", raw, 64 * i)
            msg = msg.split(b"\0")[0].decode()
            out.append({1: ("init", msg), 2: ("update", val), 3: ("end", msg)}[kind])
        return out

    def heap_check(self):
        self._cmd(OP_HEAPCHECK)
        bad, first, req, off, freed_bad, freed_req, freed_off, _ = struct.unpack("<8I", self._read(32))
        return dict(bad=bad, first=first, req=req, off=off, freed_bad=freed_bad, freed_req=freed_req, freed_off=freed_off)

    def scan_all(self, carver, line_bytes):
        self._cmd(OP_SCANALL, self.pe.symbols["lqr_carver_scan_line"], carver, line_bytes)
        lines = []
        while struct.unpack("<I", self._read(4))[0]:
            n = struct.unpack("<i", self._read(4))[0]
            lines.append((n, self._read(line_bytes)))
        return lines

    def close(self):
        if self.proc and self.proc.poll() is None:
            try:
                self._cmd(OP_QUIT)
            except Exception:
                pass
            self.proc.wait()
        self.proc = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RefCrash(RuntimeError):
    def __init__(self, msg, rc):
        super().__init__(msg)
        self.rc = rc


class RefApi:
    """stands where binding.Api stands in tests/harness.py (api.carver_class picks RefCarver)"""
    has_ext = False

    def __init__(self, cw=0x37f, poison=None, float24=False, float24_only=None):
        self.r = Runner(cw, poison, float24, float24_only)
        self.carver_class = RefCarver
        self.cw = cw

    def close(self):
        self.r.close()


class RefCarver:
    def __init__(self, api, img, init=True, delta_x=1, rigidity=0.0):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        if img.ndim == 2:
            img = img[:, :, None]
        self.api, self.r = api, api.r
        r = self.r
        self.h0, self.w0, self.ch = img.shape
        self.events, self.aux = [], []
        buf = r.alloc(max(img.nbytes, 1))         # the carver takes ownership (render.c:220-223) and g_free()s it
        r.write(buf, img.tobytes())
        self.p = r.call("lqr_carver_new", buf, self.w0, self.h0, self.ch)
        if not self.p:
            raise MemoryError("lqr_carver_new returned NULL")
        self._recording = False
        if init:
            ret = r.call("lqr_carver_init", self.p, delta_x, ("f", float(rigidity)))
            assert ret == LQR_OK, ret

    def _mask(self, mask):
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        if mask.ndim == 2:
            mask = mask[:, :, None]
        p = self.r.alloc(max(mask.nbytes, 1))
        self.r.write(p, mask.tobytes())
        return p, mask.shape

    def bias_add(self, mask, factor, x_off=0, y_off=0):
        p, (h, w, ch) = self._mask(mask)
        ret = self.r.call("lqr_carver_bias_add_rgb_area", self.p, p, int(factor), ch, w, h, x_off, y_off)
        self.r.free(p)
        return ret

    def rigmask_add(self, mask, x_off=0, y_off=0):
        p, (h, w, ch) = self._mask(mask)
        ret = self.r.call("lqr_carver_rigmask_add_rgb_area", self.p, p, ch, w, h, x_off, y_off)
        self.r.free(p)
        return ret

    def configure(self, nrg_func=2, res_order=0, switch_freq=2, enl_step=1.5, dump_vmaps=False, progress=False):
        r = self.r
        assert r.call("lqr_carver_set_energy_function_builtin", self.p, nrg_func) == LQR_OK
        r.call("lqr_carver_set_resize_order", self.p, res_order)
        if progress:
            self.set_progress_recorder()
        r.call("lqr_carver_set_side_switch_frequency", self.p, switch_freq)
        assert r.call("lqr_carver_set_enl_step", self.p, ("f", float(enl_step))) == LQR_OK
        if dump_vmaps:
            r.call("lqr_carver_set_dump_vmaps", self.p)
        return self

    def _cstr(self, s):
        p = self.r.alloc(len(s) + 1)
        self.r.write(p, s + b"\0")
        return p

    def set_progress_recorder(self):
        r = self.r
        prog = r.call("lqr_progress_new")
        r.call("lqr_progress_set_init", prog, r.cb_init)
        r.call("lqr_progress_set_update", prog, r.cb_update)
        r.call("lqr_progress_set_end", prog, r.cb_end)
        for fn, msg in (("lqr_progress_set_init_width_message", b"Resizing width..."),
                        ("lqr_progress_set_init_height_message", b"Resizing height...")):
            s = self._cstr(msg)
            r.call(fn, prog, s)
            r.free(s)
        r.call("lqr_carver_set_progress", self.p, prog)
        r.events()          # drop anything older
        self._recording = True

    def attach(self, img):
        aux = RefCarver(self.api, img, init=False)
        ret = self.r.call("lqr_carver_attach", self.p, aux.p)
        assert ret == LQR_OK, ret
        self.aux.append(aux)
        return aux

    def _pull_events(self):
        if self._recording:
            self.events += self.r.events()

    def resize(self, w1, h1):
        ret = self.r.call("lqr_carver_resize", self.p, int(w1), int(h1))
        self._pull_events()
        return ret

    def flatten(self):
        ret = self.r.call("lqr_carver_flatten", self.p)
        self._pull_events()
        return ret

    def read_scanlines(self):
        r = self.r
        W, H, ch = r.call("lqr_carver_get_width", self.p), r.call("lqr_carver_get_height", self.p), self.ch
        out = np.zeros((H, W, ch), np.uint8)
        r.call("lqr_carver_scan_reset", self.p)
        by_row = r.call("lqr_carver_scan_by_row", self.p)
        length = W if by_row else H
        lines = r.scan_all(self.p, length * ch)
        for n, raw in lines:
            buf = np.frombuffer(raw, np.uint8).reshape(length, ch)
            if by_row:
                out[n] = buf
            else:
                out[:, n] = buf
        return out, len(lines)

    def read_image(self):
        return self.read_scanlines()[0]

    def getters(self):
        r = self.r
        g = lambda n: r.calls("lqr_carver_get_" + n, self.p)      # noqa: E731
        return dict(width=g("width"), height=g("height"), channels=g("channels"), ref_width=g("ref_width"),
                    ref_height=g("ref_height"), orientation=g("orientation"), depth=g("depth"),
                    enl_step=float(np.float32(r.call("lqr_carver_get_enl_step", self.p, fp=True))))

    def _vmap_to_dict(self, v):
        r = self.r
        w, h = r.call("lqr_vmap_get_width", v), r.call("lqr_vmap_get_height", v)
        data = np.frombuffer(r.read(r.call("lqr_vmap_get_data", v), 4 * w * h), np.int32).reshape(h, w).copy()
        return dict(data=data, depth=r.calls("lqr_vmap_get_depth", v), orientation=r.calls("lqr_vmap_get_orientation", v))

    def vmap_dump(self):
        v = self.r.call("lqr_vmap_dump", self.p)
        assert v
        d = self._vmap_to_dict(v)
        self.r.call("lqr_vmap_destroy", v)
        return d

    def dumped_vmaps(self):
        r = self.r
        out = []
        lst = r.call("lqr_vmap_list_start", self.p)
        while lst:
            out.append(self._vmap_to_dict(r.call("lqr_vmap_list_current", lst)))
            lst = r.call("lqr_vmap_list_next", lst)
        return out

    def destroy(self):
        if self.p:
            self.r.call("lqr_carver_destroy", self.p)
            self.p = None
            for a in self.aux:
                a.p = None


if __name__ == "__main__":
    api = RefApi()
    r = api.r
    print("symbols:", len(r.pe.symbols), "imports:", len(r.pe.imports))
    import decimal
    decimal.getcontext().prec = 60
    for x in range(0, 65):
        got = r.call(r.c_pow, ("d", float(x)), ("d", 1.5), fp=True)
        exact = decimal.Decimal(x) ** decimal.Decimal("1.5")
        assert float(exact) == got, x
    img = (np.arange(12 * 8 * 3) * 7 % 251).astype(np.uint8).reshape(8, 12, 3)
    c = RefCarver(api, img)
    c.configure(progress=True)
    print("resize ->", c.resize(9, 8))
    print(c.getters())
    print(c.read_scanlines()[0][:, :, 0])
    print(c.vmap_dump())
    print(c.events)
    print(r.heap_check())
    c.destroy()
    print(r.heap_check(), r.info()[13:])
