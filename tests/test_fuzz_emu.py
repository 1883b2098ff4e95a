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


@pytest.mark.parametrize("seed", range(0, 24))
def test_fuzz_maxsum_emu_wide_domains(seed, oracle_built, monkeypatch):
    """The same sweep over the domains of the round-5 kernels (up to 33 values: lane grids of 16 / 64 lanes, the lane-per-edge
    class of 5..8 values, box records that overhang their tables) with their layout switches among the random flags."""
    from emu.build_emu import build
    monkeypatch.setenv("FUZZ_DOMS", "big")
    fuzz_maxsum(seed, build())
