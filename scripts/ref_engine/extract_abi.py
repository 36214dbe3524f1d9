"""Pulls the REAL liblqr prototypes and enum values out of the debug information (stabs) of the reference author's own
build (gimp-lqr-plugin.exe inside lqr-pack4win/.zip) -> tests/golden/ref/abi.json; tests/test_ref_abi.py diffs
include/lqr.h against it.  Build container only (needs /root/reference and binutils' objdump)."""
import json, os, re, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import ref_engine as R

with tempfile.TemporaryDirectory() as d:          # outside the repository
    exe = os.path.join(d, "plugin.exe")
    open(exe, "wb").write(R.exe_bytes())
    stabs = subprocess.run(["objdump", "-g", exe], capture_output=True, text=True).stdout
    syms = subprocess.run(["objdump", "-t", exe], capture_output=True, text=True).stdout

clean = lambda t: re.sub(r"\s+", " ", re.sub(r"/\*.*?\*/", "", t)).strip()      # noqa: E731
# "struct _LqrCarver *" and "enum _LqrRetVal" are how the first uses print before the typedef is known
norm = lambda t: clean(t).replace("struct _", "").replace("enum _", "").replace(" *", "*").replace("* ", "*")      # noqa: E731
functions = {}
for m in re.finditer(r"^([A-Za-z_][^\n(]*?)\b(lqr_[a-z_0-9]+) \(([^\n]*)\)\n\{", stabs, re.M):
    ret, name, args = m.groups()
    if name in functions:
        continue
    alist = []
    for a in re.split(r",(?![^(]*\))", clean(args)) if clean(args) not in ("", "void") else []:
        a = a.strip()
        t = re.match(r"(.*?)(\w+)$", a).group(1)
        alist.append(norm(t))
    functions[name] = dict(ret=norm(ret), args=alist)
enums = {}
for m in re.finditer(r"^enum _(Lqr\w+) \{ ([^}]*) \};", stabs, re.M):
    enums.setdefault(m.group(1), [x.strip() for x in m.group(2).split(",")])
addresses = {m.group(2): int(m.group(1), 16) + 0x401000 for m in re.finditer(r"0x([0-9a-f]{8}) _(lqr_\w+)$", syms, re.M)}
out = dict(source="stabs of gimp-lqr-plugin.exe (liblqr-1-0.4.1, lqr-pack4win/.zip)", functions=functions, enums=enums,
           n_symbols=len(addresses))
path = os.path.join(ROOT, "tests", "golden", "ref", "abi.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print(len(functions), "functions,", len(enums), "enums ->", path)
