"""A short run of the two random-scene campaigns (scripts/stress_parity.py, scripts/stress_oracle.py):
every walk of the hierarchy against the brute-force mode, and the GPU path against the CPU oracle
(and against the unmodified reference where its build is present), on bytes, ids and ray counters."""
import os
import sys

import pytest

import checkers

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))

pytestmark = pytest.mark.gpu


def test_walks_against_brute_force(gpu):
    import stress_parity
    assert stress_parity.run(700, 40) == 0


def test_gpu_against_oracle(gpu):
    import stress_oracle
    assert stress_oracle.run(7000, 120) == 0


@pytest.mark.skipif(not checkers.have_ref(), reason="the reference build (oracle/_ref) is not present")
def test_gpu_against_reference(gpu):
    import stress_oracle
    assert stress_oracle.run(8000, 120, against_ref=True) == 0
