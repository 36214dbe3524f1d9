"""include/lqr.h against the REAL liblqr 0.4.1 prototypes and enum values, taken from the debug information of the reference
author's own build (tests/golden/ref/abi.json, written by scripts/ref_engine/extract_abi.py from the stabs of
gimp-lqr-plugin.exe inside /root/reference/windows_installer_files/lqr-pack4win/.zip): every lqr_* function the header
declares must exist in liblqr with the same return type and the same argument types in the same order, and every enum the
header defines must list the same members with the same values."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABI = json.load(open(os.path.join(ROOT, "tests", "golden", "ref", "abi.json")))
HEADER = open(os.path.join(ROOT, "include", "lqr.h")).read()


def norm(t):
    t = re.sub(r"\bconst\b", "", t)
    return re.sub(r"\s+", " ", t).strip().replace(" *", "*").replace("* ", "*")


def header_prototypes():
    src = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    out = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z_0-9 \*]*?)\b(lqrx?_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", src):
        ret, name, args = m.groups()
        if "typedef" in ret:
            continue
        alist = []
        args = re.sub(r"\s+", " ", args).strip()
        if args not in ("", "void"):
            for a in args.split(","):
                a = re.sub(r"(\w+)\s*\[\d*\]$", r"*\1", a.strip())           # `T name[3]` is `T *name`
                alist.append(norm(re.match(r"(.*?)(\w+)$", a).group(1)))
        out[name] = dict(ret=norm(ret), args=alist)
    return out


def test_every_declared_lqr_function_has_liblqrs_own_prototype():
    protos = header_prototypes()
    declared = {n: p for n, p in protos.items() if n.startswith("lqr_")}
    assert len(declared) >= 39          # the 39 functions the plug-in binds
    for name, p in sorted(declared.items()):
        assert name in ABI["functions"], "%s is not a liblqr 0.4.1 function" % name
        real = ABI["functions"][name]
        real = dict(ret=norm(real["ret"]), args=[norm(a) for a in real["args"]])
        assert p == real, "%s: header %s, liblqr %s" % (name, p, real)


def test_enums_have_liblqrs_members_and_values():
    src = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    found = 0
    for m in re.finditer(r"typedef enum _(\w+) \{(.*?)\} (\w+);", src, re.S):
        tag, body, name = m.groups()
        members = []
        nxt = 0
        for item in [x.strip() for x in body.split(",") if x.strip()]:
            mm = re.match(r"(\w+)(?:\s*=\s*(\d+))?$", item)
            val = int(mm.group(2)) if mm.group(2) is not None else nxt
            members.append((mm.group(1), val))
            nxt = val + 1
        assert tag in ABI["enums"], tag
        assert members == [(n, i) for i, n in enumerate(ABI["enums"][tag])], (tag, members, ABI["enums"][tag])
        found += 1
    assert found >= 3      # LqrRetVal, LqrResizeOrder, LqrEnergyFuncBuiltinType


def test_callback_typedefs_return_lqrretval():
    for t in ("LqrProgressFuncInit", "LqrProgressFuncUpdate", "LqrProgressFuncEnd", "LqrVMapFunc"):
        assert re.search(r"typedef LqrRetVal \(\*%s\)" % t, HEADER), t
