"""Randomised lock-step batches (-m gpu): scripts/fuzz_batch.py with a fixed number of seeded cases -- groups of 2-9 images,
sub-batch streams 1-3, every update mode, every form of the seam round -- each image compared with the oracle's result
for it; once on plain device memory, once on recycled blocks filled with arbitrary bits (LQRHIP_POISON=r3)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("poison", ["", "r3"])
def test_seeded_batches(poison):
    env = dict(os.environ, FUZZ_COUNT="160", GPU_MAX_HW_QUEUES="16")
    if poison:
        env["LQRHIP_POISON"] = poison
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_batch.py"), "600", "7700"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " 0 failures" in r.stdout, (r.stdout[-3000:], r.stderr[-1000:])


def test_seeded_interactive_sequences():
    """scripts/fuzz_interactive.py: persistent carvers, random sequences of resizes (inside and beyond the cached map,
    both directions) and flattens; getters, image and dumped map compared with the oracle after every call"""
    env = dict(os.environ, FUZZ_COUNT="400")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_interactive.py"), "600", "7701"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " 0 failures" in r.stdout, (r.stdout[-3000:], r.stderr[-1000:])


def test_seeded_cases_with_the_plugins_other_switches():
    """scripts/fuzz_parity.py with FUZZ_EXTRAS: seam-map output, attached layers resized along, LqR-back, discard masks kept
    on enlargement, other enlargement steps -- on top of the usual seeded cases"""
    env = dict(os.environ, FUZZ_COUNT="150", FUZZ_EXTRAS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "600", "7702"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " 0 failures" in r.stdout, (r.stdout[-3000:], r.stderr[-1000:])
