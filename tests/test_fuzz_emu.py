"""A slice of the random-instance sweep of tests/fuzz_common.py (emulated engine build against the
oracles, bit for bit): domains 1..17, arities 1..4, every mode / precision / start / damping choice."""
import pytest

from fuzz_common import fuzz_maxsum, fuzz_others


@pytest.mark.parametrize("seed", range(200, 230))
def test_fuzz_maxsum_emu(seed, oracle_built):
    from emu.build_emu import build
    fuzz_maxsum(seed, build())


@pytest.mark.parametrize("seed", list(range(0, 60)) + list(range(200, 225)))  # (0..59: the seeds the GPU twin runs)
def test_fuzz_amaxsum_dsa_mgm_emu(seed, oracle_built):
    from emu.build_emu import build
    fuzz_others(seed, build())
