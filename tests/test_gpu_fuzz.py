"""The random-instance sweep of tests/fuzz_common.py on the GPU: every engine of the library against
its oracle, bit for bit, on instances nobody picked (domains 1..17, arities 1..4, every mode /
precision / start / damping choice, random layout flags)."""
import pytest

from fuzz_common import fuzz_maxsum, fuzz_others

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(0, 80))
def test_fuzz_maxsum(seed, oracle_built):
    fuzz_maxsum(seed, None)


@pytest.mark.parametrize("seed", range(0, 60))
def test_fuzz_amaxsum_dsa_mgm(seed, oracle_built):
    fuzz_others(seed, None)


@pytest.mark.parametrize("seed", range(0, 40))
def test_fuzz_maxsum_wide_domains(seed, oracle_built, monkeypatch):
    """Domains up to 33 values (round 5: lane-grid factor kernels, the lane-per-edge variable class of 5..8 values, box records
    overhanging their tables), their layout switches among the random flags."""
    monkeypatch.setenv("FUZZ_DOMS", "big")
    fuzz_maxsum(seed, None)
