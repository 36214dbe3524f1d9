"""Register and scratch budgets the design depends on, checked at BUILD time (CPU test, no GPU needed).

The persistent kernels spin on their neighbours and are sized at run time from the occupancy query, and the streaming carve
gets its bandwidth from waves per SIMD: a harmless-looking edit that lets the scheduler take a kernel over a register
boundary changes all of that "without a word" (DESIGN.md 4.15 lesson vii: k_band_tiles went from 225 to 262 registers and
the residency bound from 768 to 256 workgroups; round 2's carve at 142 VGPRs ran 3 waves per SIMD instead of 5).  This test
reads the shipped library's code-object metadata (tests/kernel_meta.py) and fails the build instead of the user.

Checked red on a deliberately fattened build: `make -C gimp-lqr-plugin_amd EXTRA=-DCG=6 BUILD=/tmp/fat OUT=/tmp/fat/lib.so`
(the carve's group size: 104 VGPRs) and `LQR_BUDGET_LIB=/tmp/fat/lib.so pytest tests/test_kernel_budgets.py` -> k_carve
over budget; test_checker_is_red_on_a_fattened_kernel does the same on doctored metadata in every run."""
import copy
import functools
import os
import re

import pytest

import kernel_meta as KM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("LQR_BUDGET_LIB") or os.path.join(ROOT, "gimp-lqr-plugin_amd", "liblqr-hip.so")

# (regular expression on the demangled name, max VGPRs + AGPRs, max scratch bytes, max spilled VGPRs, why)
BUDGETS = [
    (r"^k_carve$", 96, 0, 0, "5 waves per SIMD: the carve's 0.4 of the HBM roof next to the chain kernels (DESIGN 4.9, 4.14)"),
    (r"^k_band_update_tw<4, ", 256, 0, 0, "8 waves per workgroup = 2 per SIMD; it claims all 256 registers on purpose (no sibling kernel's wave beside it); no scratch in the row loop"),
    (r"^k_band_levels<", 256, 0, 0, "2 waves per SIMD (amdgpu_waves_per_eu(2, 2)): the residency bound the slot count is taken from; one site each for loads, rows and stores or 330 VGPRs spill (DESIGN 4.16)"),
    (r"^k_dp_tile_p<[24], (true|false), (true|false), false, 1, false, (16|24)>$", 128, 0, 0, "E5, plain: 4 waves per SIMD"),
    (r"^k_dp_tile_p<2, (true|false), (true|false), true, 1, false, (16|24)>$", 232, 0, 0, "E9 full width, 32-row block staged in registers: 2 waves per SIMD"),
    (r"^k_dp_tile_p<4, (true|false), (true|false), true, 1, false, 16>$", 216, 0, 0, "E9 full width, 4 px per lane: 2 waves per SIMD"),
    (r"^k_dp_tile_p<2, .*, [1234], (true|false), 16>$", 272, 0, 0, "general instantiations (delta_x 2..4, rigidity mask): at least one workgroup per SIMD pair, no scratch"),
    (r"^k_dp_tile_p<2, (true|false), true, (true|false), ([5-9]|10), (true|false), 16>$", 256, 0, 0, "delta_x 5 .. 10 (round 6): 3 .. 6 staged rows, 11 .. 21 candidates per pixel; two workgroups per SIMD pair, no scratch"),
    (r"^k_vp_maps<", 72, 0, 0, "the map kernel is LDS-latency-bound: 7 workgroups per CU (21 KB of LDS each) = 7 waves per SIMD; the staging loads of a thread (21 dwords) are in flight together"),
    (r"^k_vp_solve<", 160, 0, 0, "one workgroup per image; a stage's loads are all issued before the first is stored (12 x 16 bytes in registers)"),
    (r"^k_vpath1<1>$", 192, 0, 0, "one wave chases, 2 waves per SIMD of the 4-wave workgroup"),
    (r"^k_vpath1<[234567]>$", 128, 0, 0, "shorter chunks"),
    (r"^k_emap_update<\d, 12>$", 64, 0, 0, "delta_x <= 2: 8 waves per SIMD"),
    (r"^k_dp_tile<", 96, 0, 0, "one wave per tile, 5 per SIMD"),
]
# kernels that are allowed to use scratch at all (slow paths for rows wider than 8192 px / known, outside the row loops)
SCRATCH_OK = (r"^k_dp_sweep<16, ", r"^k_dp_sweep<8, true, (1024|256)>$")


def violations(meta):
    bad = []
    seen = set()
    for name, d in sorted(meta.items()):
        regs = d["vgpr_count"] + d["agpr_count"]
        if d["private_segment_fixed_size"] and not any(re.search(p, name) for p in SCRATCH_OK):
            bad.append("%s: %d bytes of scratch (spills: %d VGPRs, %d SGPRs); only %s may use scratch" % (
                name, d["private_segment_fixed_size"], d["vgpr_spill_count"], d["sgpr_spill_count"], ", ".join(SCRATCH_OK)))
        for pat, max_regs, max_scratch, max_spill, why in BUDGETS:
            if re.search(pat, name):
                seen.add(pat)
                if regs > max_regs or d["private_segment_fixed_size"] > max_scratch or d["vgpr_spill_count"] > max_spill:
                    bad.append("%s: %d VGPRs (+AGPRs), %d B scratch, %d spilled VGPRs -- budget %d / %d / %d: %s" % (
                        name, regs, d["private_segment_fixed_size"], d["vgpr_spill_count"], max_regs, max_scratch, max_spill, why))
                break
    for pat, *_ in BUDGETS:
        if pat not in seen:
            bad.append("no kernel matches budget pattern %s (renamed? the budget must follow it)" % pat)
    return bad


@pytest.fixture(scope="module")
def meta():
    if not KM.tools_available():
        pytest.skip("no LLVM object tools on this machine")
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    return KM.kernels(LIB)


def test_every_kernel_is_within_its_budget(meta):
    assert len(meta) > 100, "expected the whole kernel set, found %d kernels" % len(meta)
    bad = violations(meta)
    assert not bad, "\n".join(bad)


def test_checker_is_red_on_a_fattened_kernel(meta):
    """the same check on doctored metadata: the carve at round 2's 142 VGPRs, k_band_levels at 262 (one wave per SIMD), scratch in
    the trapezoid band kernel -- each must be reported"""
    fat = copy.deepcopy(meta)
    fat["k_carve"]["vgpr_count"] = 142
    fat["k_band_levels<false, false, 1, false>"]["vgpr_count"] = 262
    fat["k_band_update_tw<4, true, false>"]["private_segment_fixed_size"] = 64
    bad = violations(fat)
    assert len(bad) >= 3 and any(b.startswith("k_carve:") for b in bad) and any("k_band_levels<false, false, 1, false>" in b for b in bad) \
        and any("k_band_update_tw<4, true, false>" in b for b in bad), bad


# ---- what the disassembly must (not) contain: findings of round 5 that a compile can silently undo
# (regular expression on the demangled name, mnemonic, max count, why)
ISA_LIMITS = [
    (r"^k_band_levels<(true|false), false, 1, false>$", "s_nop", 40,
     "the 32-row loop with its compares in free SGPR pairs (13 hazard nops in the kernel); when the scalar registers run out the "
     "register allocator serialises every compare through VCC: 4 nops per row, 141 in the kernel, 261 -> 284 us at 8 images "
     "(DESIGN.md 4.16: any edit of the kernel can flip it; measured with seven variants)"),
    (r"^k_band_levels<(true|false), true, 1, false>$", "s_nop", 80,
     "the rigidity instantiations get the same row schedule by NOT holding the per-lane predicates as scalar-register pairs (LEAN in "
     "k_levels.hip): 50 nops, of which 32 are single wait states between the prefetch loads; 138 with the VCC-serialised rows"),
    (r"^k_band_update_tw<4, ", "v_readfirstlane_b32", 40,
     "plane pointers in scalar registers (uni_ptr): from the descriptor's vector loads they arrive in VGPRs and every row paid "
     "4 v_readfirstlane + hazard nops (202 in the kernel)"),
    (r"^k_band_update_tw<4, ", "s_nop", 40, "as above (106 before)"),
    (r"^k_dp_tile_p<2, (true|false), false, true, 1, false, (16|24)>$", "s_nop", 40, "the 32-row block loop, as k_band_levels (10 now)"),
    (r"^k_(band|dp_tile|dp_sweep|vpath|carve|emap)", "flat_load_dword", 0,
     "an LDS flag read through a generic pointer (volatile cast of a __shared__ variable inside a lambda): FLAT + s_waitcnt "
     "vmcnt(0) drains the wave's prefetch and waits for its write-through stores -- use LDS_FLAG (lqr_common.h)"),
    (r"^k_(band|dp_tile|dp_sweep|vpath|carve|emap)", "flat_store_dword", 0, "as above (the one-off kernels of k_oneoff.hip take generic pointers and may)"),
]


@functools.lru_cache(maxsize=None)
def isa_counts():
    return KM.instruction_counts(LIB)


def isa_violations(counts):
    bad = []
    seen = set()
    for name, d in sorted(counts.items()):
        for pat, mn, limit, why in ISA_LIMITS:
            if re.search(pat, name):
                seen.add((pat, mn))
                if d.get(mn, 0) > limit:
                    bad.append("%s: %d x %s, limit %d: %s" % (name, d[mn], mn, limit, why))
    for pat, mn, *_ in ISA_LIMITS:
        if (pat, mn) not in seen:
            bad.append("no kernel matches ISA pattern %s (renamed? the limit must follow it)" % pat)
    return bad


# the instruction-schedule limits (s_nop, v_readfirstlane) describe what THIS compiler makes of the row loops; another hipcc release
# may schedule differently without anything being wrong -- then they are reported, not failed.  The FLAT limits hold for any.
RECORDED_COMPILER = "roc-7.2.0"


def compiler_is_the_recorded_one():
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    try:
        return RECORDED_COMPILER in subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
    except OSError:
        return False


def test_row_loops_keep_their_schedule_and_no_flag_goes_through_flat(meta):
    counts = isa_counts()
    assert len(counts) > 100, "expected the whole kernel set, found %d kernels" % len(counts)
    bad = isa_violations(counts)
    flat = [b for b in bad if " x flat_" in b or "no kernel matches" in b]
    sched = [b for b in bad if b not in flat]
    assert not flat, "\n".join(flat)
    if sched and not compiler_is_the_recorded_one():
        pytest.xfail("schedule limits recorded for %s; this compiler: %s" % (RECORDED_COMPILER, "; ".join(sched)))
    assert not sched, "\n".join(sched)


def test_isa_checker_is_red_on_doctored_counts(meta):
    fat = copy.deepcopy(isa_counts())
    fat["k_band_levels<false, false, 1, false>"]["s_nop"] = 141
    fat["k_dp_tile_p<2, true, false, true, 1, false, 16>"]["flat_load_dword"] = 3
    fat["k_band_update_tw<4, false, false>"]["v_readfirstlane_b32"] = 202
    bad = isa_violations(fat)
    assert len(bad) == 3 and "k_band_levels" in bad[0] and "k_band_update_tw" in bad[1] and "flat_load_dword" in bad[2], bad
