"""Register / scratch / LDS figures of every kernel in a built liblqr-hip.so, read from the code objects' metadata
(the .hip_fatbin section holds one clang offload bundle per translation unit; each gfx950 code object carries an
amdhsa metadata note with .vgpr_count, .sgpr_count, .vgpr_spill_count, .private_segment_fixed_size, ... per kernel).
Used by tests/test_kernel_budgets.py; `python tests/kernel_meta.py [lib.so]` prints the table."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
KEYS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
        "group_segment_fixed_size", "max_flat_workgroup_size")


def tools_available():
    return all(os.path.exists(os.path.join(LLVM_BIN, t)) for t in ("clang-offload-bundler", "llvm-readelf", "llvm-objdump")) and shutil.which("objcopy") and shutil.which("c++filt")


def kernels(so_path):
    """{demangled kernel name (without the parameter list): {key: int}}"""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so_path, fat])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        assert starts, "no offload bundle in %s" % so_path
        mangled = []
        for i, a in enumerate(starts):
            b = starts[i + 1] if i + 1 < len(starts) else len(blob)
            piece, co = os.path.join(td, "b%d.bin" % i), os.path.join(td, "b%d.co" % i)
            open(piece, "wb").write(blob[a:b])
            subprocess.check_call([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + piece, "--output=" + co])
            if os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
            for blk in re.split(r"\n  - \.agpr_count", notes)[1:]:
                blk = ".agpr_count" + blk
                d = {}
                for k in KEYS:
                    m = re.search(r"\.%s:\s+(\d+)" % k, blk)
                    d[k] = int(m.group(1)) if m else 0
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                mangled.append((name, d))
        names = subprocess.run(["c++filt"], input="\n".join(n for n, _ in mangled), capture_output=True, text=True, check=True).stdout.split("\n")
        for (_, d), n in zip(mangled, names):
            n = re.sub(r"^void ", "", n)
            n = re.sub(r"\(.*", "", n)
            out[n] = d
    return out


def instruction_counts(so_path, mnemonics=("s_nop", "v_readfirstlane_b32", "flat_load_dword", "flat_store_dword")):
    """{demangled kernel name: {mnemonic: count}} from the disassembly of every gfx950 code object in the library"""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so_path, fat])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        counts = {}
        for i, a in enumerate(starts):
            b = starts[i + 1] if i + 1 < len(starts) else len(blob)
            piece, co = os.path.join(td, "b%d.bin" % i), os.path.join(td, "b%d.co" % i)
            open(piece, "wb").write(blob[a:b])
            subprocess.check_call([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + piece, "--output=" + co])
            if os.path.getsize(co) == 0:
                continue
            dis = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
            cur = None
            for line in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = counts.setdefault(m.group(1), dict.fromkeys(mnemonics, 0))
                    continue
                if cur is None:
                    continue
                t = line.split()
                if not t:
                    continue
                if t[0] in cur:
                    cur[t[0]] += 1
                # (every width of FLAT access counts -- flat_load_dwordx2, _ubyte, _short ... -- under the dword key's family)
                elif t[0].startswith("flat_load_") and "flat_load_dword" in cur:
                    cur["flat_load_dword"] += 1
                elif t[0].startswith("flat_store_") and "flat_store_dword" in cur:
                    cur["flat_store_dword"] += 1
        mangled = sorted(counts)
        names = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True, check=True).stdout.split("\n")
        for mn, n in zip(mangled, names):
            n = re.sub(r"^void ", "", n)
            n = re.sub(r"\(.*", "", n)
            out[n] = counts[mn]
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "gimp-lqr-plugin_amd", "liblqr-hip.so")
    for n, d in sorted(kernels(so).items()):
        print("%-60s vgpr %3d agpr %2d sgpr %3d spill v %3d s %3d scratch %4d lds %6d" % (
            n[:60], d["vgpr_count"], d["agpr_count"], d["sgpr_count"], d["vgpr_spill_count"], d["sgpr_spill_count"],
            d["private_segment_fixed_size"], d["group_segment_fixed_size"]))
