"""Nothing the engine computes may depend on what a recycled device block held before (-m gpu).

The engine's allocator hands freed blocks out again (csrc/lqr_shim.hip, pool_alloc), so a kernel that reads a cell nothing
wrote yet sees another image's planes -- and passes every test that runs a case on a fresh process.  LQRHIP_POISON (read
once, when the library is loaded: hence the child processes) fills every block handed out with a pattern first:

  r3  arbitrary bits: signalling NaNs, infinities, huge negatives -- the fill that found the stale-NaN hole in the tiled
      update's keep rule (addendum 2), cases 222 / 564 / 660 of seed 778
  r1  floats in [0, 100): plausible cumulative energies
  255 every float a quiet NaN, every back pointer -1

Each child runs seeded cases of tests/fuzz_cases.py against the oracle (scripts/fuzz_parity.py)."""
import os
import subprocess
import sys

import pytest

import fuzz_common as FC

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_fuzz(poison, seconds, seed, extra=(), only=None, count=0):
    if count:           # count-bounded, every case asked for must have run (tests/fuzz_common.py)
        return FC.run_script("fuzz_parity.py", [0, seed, *extra], dict(LQRHIP_POISON=poison), count)
    env = dict(os.environ, LQRHIP_POISON=poison)
    if only:
        env["FUZZ_ONLY"] = only
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), str(seconds), str(seed), *extra],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, "exit %d\n--- stdout ---\n%s\n--- stderr ---\n%s" % (r.returncode, r.stdout, r.stderr)
    return r.stdout


def test_the_cases_that_found_it():
    out = run_fuzz("r3", 999, 778, only="222,564,660")
    assert out.count("ok   case") == 3, out[-2000:]


@pytest.mark.parametrize("poison", ["r3", "r1", "255"])
def test_seeded_cases_on_poisoned_blocks(poison):
    out = run_fuzz(poison, 600, 4242, count=200)
    assert " 0 failures" in out, out[-2000:]


def test_general_kernels_on_poisoned_blocks():
    out = run_fuzz("r3", 600, 4243, extra=("0", "general"), count=300)
    assert " 0 failures" in out, out[-2000:]
