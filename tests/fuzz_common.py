"""Shared by the randomised parity scripts (scripts/fuzz_*.py) and the tests that run them as child processes.

Two things a fuzz run must not be able to hide (VERDICT r4, weak point 1):
  * how many cases it ran: with FUZZ_COUNT set a run is COUNT-bounded only -- no wall-clock exit, so a slow box cannot run
    fewer cases than asked and still print "0 failures"; the summary says `cases run / cases asked` and the tests assert it;
  * what a failure was: a result that differs from the oracle's (MISMATCH), a persistent kernel's spin wait that gave up
    (TIMEOUT, DEVERR_TILE_TIMEOUT), a failed activity prediction (DEVERR), any other error return (ERROR) -- each FAIL line
    carries the kind, the engine's last error string, its fault counters and k_band_levels' event counters.
"""
import ctypes
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KINDS = ("mismatch", "timeout", "deverr", "error")


class Budget:
    """FUZZ_COUNT=n: exactly n cases, however long they take; otherwise `seconds` of wall clock."""

    def __init__(self, seconds):
        self.count = int(os.environ.get("FUZZ_COUNT", "0"))
        self.t_end = time.time() + seconds

    def more(self, n):
        return n < self.count if self.count else time.time() < self.t_end

    @property
    def asked(self):
        return self.count


class Failures:
    def __init__(self, lib=None):
        self.lib = lib
        self.by_kind = dict.fromkeys(KINDS, 0)

    @property
    def total(self):
        return sum(self.by_kind.values())

    def record(self, case, what, ex):
        """classify, print ONE line with everything needed to attribute it, count"""
        msg = str(ex)
        err = ""
        stats = ""
        if self.lib is not None:
            try:
                self.lib.lqrhip_last_error.restype = ctypes.c_char_p
                err = (self.lib.lqrhip_last_error() or b"").decode(errors="replace")
            except Exception:
                pass
            try:
                out = (ctypes.c_ulonglong * 8)()
                self.lib.lqrhip_fault_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
                if self.lib.lqrhip_fault_stats(out, 0) == 0:
                    stats = " faults[timeouts,predictions,seamlog,levels,rolled_back,injected,redone]=%s" % [int(x) for x in out[:7]]
            except Exception:
                pass
            try:
                out = (ctypes.c_ulonglong * 8)()
                self.lib.lqrhip_band_levels_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
                if self.lib.lqrhip_band_levels_stats(out, 0) == 0:
                    stats += " lv_stats[collisions,sync_loads,processed,idle]=%s" % [int(x) for x in out[:4]]
            except Exception:
                pass
        if "never became resident" in err or "never became resident" in msg:
            kind = "timeout"
        elif "prediction failed" in err or "prediction failed" in msg:
            kind = "deverr"
        elif "differ" in msg or "seam maps" in msg or "pixels" in msg or "DP planes" in msg or "getters" in msg or "image at" in msg or "map at" in msg:
            kind = "mismatch"
        else:
            kind = "error"
        self.by_kind[kind] += 1
        print("FAIL[%s] case %s %s :: %s :: last_error=%r%s" % (kind.upper(), case, what, msg[:300], err[:200], stats), flush=True)
        return kind


def summary(name, n, budget, fails, seed, extra=""):
    asked = budget.asked
    print("%s: %d cases run / %s asked%s, %d failures (%s), seed %d" % (
        name, n, asked if asked else "time-bounded", extra, fails.total,
        ", ".join("%s %d" % (k, fails.by_kind[k]) for k in KINDS), seed), flush=True)


SUMMARY_RE = re.compile(r"(\d+) cases run / (\d+|time-bounded) asked.*?, (\d+) failures \(mismatch (\d+), timeout (\d+), deverr (\d+), error (\d+)\)")


def run_script(script, args, env_extra, expect_cases, timeout=2400):
    """Run scripts/<script> as a child process with FUZZ_COUNT=expect_cases and assert that it ran exactly that many cases
    without a failure.  The assertion message holds the child's whole stdout and stderr, and says which kind of failure."""
    env = dict(os.environ, FUZZ_COUNT=str(expect_cases))
    env.update(env_extra)
    cmd = [sys.executable, os.path.join(ROOT, "scripts", script)] + [str(a) for a in args]
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired as ex:
        def txt(b):
            return b.decode(errors="replace") if isinstance(b, bytes) else (b or "")
        raise AssertionError("CHILD TIME-OUT after %d s (not a mismatch): %s\n--- stdout ---\n%s\n--- stderr ---\n%s" % (
            timeout, " ".join(cmd), txt(ex.stdout), txt(ex.stderr)))
    report = "%s -> exit %d\n--- stdout ---\n%s\n--- stderr ---\n%s" % (" ".join(cmd), r.returncode, r.stdout, r.stderr)
    m = SUMMARY_RE.search(r.stdout)
    assert m, "no summary line (child crashed?): " + report
    ran, asked, nfail = int(m.group(1)), m.group(2), int(m.group(3))
    kinds = dict(zip(KINDS, (int(m.group(i)) for i in range(4, 8))))
    assert nfail == 0 and r.returncode == 0, "%d failing cases %s: %s" % (nfail, kinds, report)
    assert asked == str(expect_cases) and ran == expect_cases, "ran %d of %s cases asked (%d expected): %s" % (ran, asked, expect_cases, report)
    return r.stdout
