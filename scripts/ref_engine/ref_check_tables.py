"""Facts for oracle/REF_CHECK.md pulled out of the genuine build mechanically: per engine function its address range,
the liblqr source lines it was compiled from (stabs), the functions it calls and the floating-point constants it loads
from .rdata.  Build container only; prints markdown."""
import os, re, struct, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_engine as R

FUNCS = ["lqr_carver_new_common", "lqr_carver_init", "lqr_carver_init_energy_related", "lqr_carver_build_maps", "lqr_carver_build_emap", "lqr_carver_compute_e",
         "lqr_carver_build_mmap", "lqr_carver_build_vpath", "lqr_carver_update_vsmap", "lqr_carver_carve", "lqr_carver_update_emap",
         "lqr_carver_update_mmap", "lqr_carver_finish_vsmap", "lqr_carver_build_vsmap", "lqr_carver_inflate", "lqr_carver_flatten",
         "lqr_carver_transpose", "lqr_carver_resize_width", "lqr_carver_resize_height", "lqr_carver_resize", "lqr_carver_set_width",
         "lqr_carver_scan_line_ext", "lqr_carver_scan_ext", "lqr_vmap_dump", "lqr_vmap_internal_dump", "lqr_carver_bias_add_xy", "lqr_carver_bias_add_rgb_area",
         "lqr_carver_rigmask_add_xy", "lqr_carver_rigmask_add_rgb_area", "lqr_pixel_get_rgbcol", "lqr_pixel_get_norm", "lqr_carver_read_brightness_std",
         "lqr_carver_read_luma_std", "lqr_carver_generate_rcache_bright", "lqr_carver_generate_rcache_luma", "lqr_rwindow_fill_std",
         "lqr_energy_builtin_grad_all", "lqr_grad_norm", "lqr_grad_sumabs", "lqr_grad_xabs", "lqr_carver_set_enl_step", "lqr_progress_new"]
pe = R.PE(R.exe_bytes())
with tempfile.TemporaryDirectory() as d:
    exe = os.path.join(d, "p.exe"); open(exe, "wb").write(pe.data)
    dis = subprocess.run(["objdump", "-d", exe], capture_output=True, text=True).stdout
    stabs = subprocess.run(["objdump", "-g", exe], capture_output=True, text=True).stdout
rdata = next(s for s in pe.sections if s[0] == ".rdata")
rd_lo = pe.image_base + rdata[1]; rd_hi = rd_lo + rdata[2]
def rd(addr, n): o = rdata[3] + addr - rd_lo; return pe.data[o:o + n]
lines = {}
for m in re.finditer(r"/\* file (\S+) line (\d+) addr (0x[0-9a-f]+) \*/", stabs):
    lines[int(m.group(3), 16)] = (os.path.basename(m.group(1)), int(m.group(2)))
addr_sorted = sorted((a, n) for n, a in pe.symbols.items() if n.startswith("lqr_") or n.startswith("_"))
blocks = re.split(r"\n(?=[0-9a-f]{8} <)", dis)
body = {}
for b in blocks:
    m = re.match(r"([0-9a-f]{8}) <_(\w+)>:", b)
    if m: body[m.group(2)] = (int(m.group(1), 16), b)
print("| function | address | source (liblqr-1-0.4.1/lqr/) | calls | .rdata constants loaded |")
print("|---|---|---|---|---|")
for f in FUNCS:
    if f not in body: continue
    start, b = body[f]
    insn = re.findall(r"^\s+([0-9a-f]+):\t[^\t]*\t(.*)$", b, re.M)
    end = int(insn[-1][0], 16)
    ls = [lines[a] for a in range(start, end + 1) if a in lines]
    files = sorted(set(x[0] for x in ls))
    src = ", ".join("%s:%d-%d" % (fn, min(l for g, l in ls if g == fn), max(l for g, l in ls if g == fn)) for fn in files if fn.endswith(".c"))
    calls = sorted(set(re.findall(r"call\s+[0-9a-f]+ <_(\w+)>", b)))
    consts = []
    for op, a in re.findall(r"(fld[sl]|fmul[sl]|fadd[sl]|fdiv[sl]|fdivr[sl]|fsub[sl]|fcom[sl]|fcomp[sl])\s+0x(41[c-f][0-9a-f]{3})\b", b):
        a = int(a, 16)
        v = struct.unpack("<f", rd(a, 4))[0] if op.endswith("s") else struct.unpack("<d", rd(a, 8))[0]
        consts.append("%#x=%s%r" % (a, "(float)" if op.endswith("s") else "(double)", v))
    consts = sorted(set(consts))
    print("| `%s` | %#x-%#x | %s | %s | %s |" % (f, start, end, src, ", ".join(c.replace("lqr_carver_", "").replace("lqr_", "") for c in calls) or "-", "; ".join(consts) or "-"))
