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
    (r"^k_band_tiles<", 256, 32, 8, "2 waves per SIMD (amdgpu_waves_per_eu(2, 2)): the residency bound of 768 workgroups; the few spills sit outside the row loop"),
    (r"^k_dp_tile_p<[24], (true|false), (true|false), false, 1, false>$", 128, 0, 0, "E5, plain: 4 waves per SIMD"),
    (r"^k_dp_tile_p<2, (true|false), (true|false), true, 1, false>$", 232, 0, 0, "E9 full width, 32-row block staged in registers: 2 waves per SIMD"),
    (r"^k_dp_tile_p<4, (true|false), (true|false), true, 1, false>$", 216, 0, 0, "E9 full width, 4 px per lane: 2 waves per SIMD"),
    (r"^k_dp_tile_p<2, .*, [1234], (true|false)>$", 272, 0, 0, "general instantiations (delta_x 2..4, rigidity mask): at least one workgroup per SIMD pair, no scratch"),
    (r"^k_vpath1<1>$", 192, 0, 0, "one wave chases, 2 waves per SIMD of the 4-wave workgroup"),
    (r"^k_vpath1<[234]>$", 128, 0, 0, "shorter chunks"),
    (r"^k_emap_update<\d, 12>$", 64, 0, 0, "delta_x <= 2: 8 waves per SIMD"),
    (r"^k_dp_tile<", 96, 0, 0, "one wave per tile, 5 per SIMD"),
]
# kernels that are allowed to use scratch at all (slow paths for rows wider than 8192 px / known, outside the row loops)
SCRATCH_OK = (r"^k_dp_sweep<16, ", r"^k_dp_sweep<8, true>$", r"^k_band_tiles<")


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
    """the same check on doctored metadata: the carve at round 2's 142 VGPRs, k_band_tiles at lesson (vii)'s 262, scratch in
    the trapezoid band kernel -- each must be reported"""
    fat = copy.deepcopy(meta)
    fat["k_carve"]["vgpr_count"] = 142
    fat["k_band_tiles<false, false>"]["vgpr_count"] = 262
    fat["k_band_update_tw<4, true, false>"]["private_segment_fixed_size"] = 64
    bad = violations(fat)
    assert len(bad) >= 3 and any(b.startswith("k_carve:") for b in bad) and any("k_band_tiles<false, false>" in b for b in bad) \
        and any("k_band_update_tw<4, true, false>" in b for b in bad), bad
