"""MGM on the emulated engine build (the very same mgm.hip, g++ against the fake HIP runtime)
against the oracle, bit for bit -- the CPU twin of tests/test_gpu_mgm.py."""
import pytest

from mgm_common import compare_mgm, mgm_cases
from pydcop_amd.graph import Params


@pytest.mark.parametrize("case", mgm_cases(), ids=lambda c: c[0])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_mgm_emu_bit_exact_vs_oracle(case, dtype, oracle_built):
    from emu.build_emu import build
    from oracle.mgm_oracle import OracleMgm
    name, make, kw = case
    compare_mgm(OracleMgm, make(), Params(dtype=dtype, **kw), lib_path=build())


@pytest.mark.parametrize("case", mgm_cases()[:5], ids=lambda c: c[0])
def test_mgm_emu_csr_walk_kernels(case, oracle_built, monkeypatch):
    """MAXSUM_LOCAL_SEARCH_GENERIC=1: the CSR-walk kernels (what domains of more than 32 values
    run on) on the instances the slot kernels were just checked on."""
    from emu.build_emu import build
    from oracle.mgm_oracle import OracleMgm
    monkeypatch.setenv("MAXSUM_LOCAL_SEARCH_GENERIC", "1")
    name, make, kw = case
    compare_mgm(OracleMgm, make(), Params(**kw), lib_path=build(), steps=(0, 1, 3, 6))


@pytest.mark.parametrize("case", mgm_cases()[:6], ids=lambda c: c[0])
def test_mgm_emu_slot_kernels_everywhere(case, oracle_built, monkeypatch):
    """MAXSUM_LOCAL_SEARCH_GENERIC=2: the thread-per-variable slot kernels also for the variables
    the packed (lane per constraint) kernels take by default."""
    from emu.build_emu import build
    from oracle.mgm_oracle import OracleMgm
    monkeypatch.setenv("MAXSUM_LOCAL_SEARCH_GENERIC", "2")
    name, make, kw = case
    compare_mgm(OracleMgm, make(), Params(**kw), lib_path=build(), steps=(0, 1, 3, 6))


@pytest.mark.parametrize("case", [c for c in mgm_cases() if c[0].startswith(("meeting", "mixed"))], ids=lambda c: c[0])
def test_mgm_emu_strided_slots_without_the_row_view(case, oracle_built, monkeypatch):
    """MAXSUM_LOCAL_SEARCH_ROWS=0: no private row copies (local_search.h, Slots::rows) -- the strided slot reads, the
    path of instances whose copies exceed the budget."""
    from emu.build_emu import build
    from oracle.mgm_oracle import OracleMgm
    monkeypatch.setenv("MAXSUM_LOCAL_SEARCH_ROWS", "0")
    name, make, kw = case
    compare_mgm(OracleMgm, make(), Params(**kw), lib_path=build(), steps=(0, 1, 3, 6))
