"""A short run of the random-scene campaigns (scripts/stress_parity.py, scripts/stress_long.py, scripts/stress_oracle.py):
every walk of the hierarchy against the brute-force mode, the brute-force mode against the default mode on long ray lists, and
the GPU path against the CPU oracle (and against the unmodified reference where its build is present), on bytes, ids and ray counters."""
import os
import sys

import pytest

import checkers

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))

pytestmark = pytest.mark.gpu


# The scenes are fixed by their seeds (nothing here depends on time or on the box): the same scenes every run.
def test_walks_against_brute_force(gpu):
    """Seeds 700-739: 40 scenes, each in the default mode, both forced walks and the brute-force mode as written and in exact stages, against the brute-force mode."""
    import stress_parity
    assert stress_parity.run(700, 40) == 0, "seeds 700-739, see the MISMATCH lines above"


def test_brute_force_against_default_on_long_ray_lists(gpu):
    """Seeds 100-111: 12 scenes of 15-33 M samples and up to eight lights, the brute-force mode twice (and in exact stages) against the default mode."""
    import stress_long
    assert stress_long.run(100, 12) == 0, "seeds 100-111, see the MISMATCH lines above"


def test_gpu_against_oracle(gpu):
    """Seeds 7000-7119: 120 small scenes against the CPU oracle (bytes, ids, ray counters)."""
    import stress_oracle
    assert stress_oracle.run(7000, 120) == 0, "seeds 7000-7119, see the MISMATCH lines above"


@pytest.mark.skipif(not checkers.have_ref(), reason="the reference build (oracle/_ref) is not present")
def test_gpu_against_reference(gpu):
    """Seeds 8000-8119: 120 small scenes against the unmodified reference (bytes)."""
    import stress_oracle
    assert stress_oracle.run(8000, 120, against_ref=True) == 0, "seeds 8000-8119, see the MISMATCH lines above"
