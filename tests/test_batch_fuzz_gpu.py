"""Randomised lock-step batches (-m gpu): scripts/fuzz_batch.py with a fixed number of seeded cases -- groups of 2-9 images,
sub-batch streams 1-3, every update mode, every form of the seam round -- each image compared with the oracle's result
for it; once on plain device memory, once on recycled blocks filled with arbitrary bits (LQRHIP_POISON=r3).

The children are COUNT-bounded (no wall-clock exit), the tests assert that every case asked for ran, and a failure's message
holds the child's whole output with the kind of each failing case (mismatch / time-out / device error): tests/fuzz_common.py."""
import pytest

import fuzz_common as FC

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("poison", ["", "r3"])
def test_seeded_batches(poison):
    FC.run_script("fuzz_batch.py", [0, 7700], dict(GPU_MAX_HW_QUEUES="16", **({"LQRHIP_POISON": poison} if poison else {})), 160)


def test_seeded_interactive_sequences():
    """scripts/fuzz_interactive.py: persistent carvers, random sequences of resizes (inside and beyond the cached map,
    both directions) and flattens; getters, image and dumped map compared with the oracle after every call"""
    FC.run_script("fuzz_interactive.py", [0, 7701], {}, 400)


def test_seeded_cases_with_the_plugins_other_switches():
    """scripts/fuzz_parity.py with FUZZ_EXTRAS: seam-map output, attached layers resized along, LqR-back, discard masks kept
    on enlargement, other enlargement steps -- on top of the usual seeded cases"""
    FC.run_script("fuzz_parity.py", [0, 7702], dict(FUZZ_EXTRAS="1"), 150)
