"""DSA on the emulated engine build (the very same dsa.hip, g++ against the fake HIP runtime)
against the oracle, bit for bit -- the CPU twin of tests/test_gpu_dsa.py."""
import pytest

from dsa_common import compare_dsa, dsa_cases
from pydcop_amd.graph import Params


@pytest.mark.parametrize("case", dsa_cases(k=4), ids=lambda c: c[0])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_dsa_emu_bit_exact_vs_oracle(case, dtype, oracle_built):
    from emu.build_emu import build
    from oracle.dsa_oracle import OracleDsa
    name, make, kw, dsa_kw = case
    compare_dsa(OracleDsa, make(), Params(dtype=dtype, **kw), dsa_kw, lib_path=build(), steps=(0, 1, 1, 3, 10))


@pytest.mark.parametrize("case", dsa_cases(k=4)[:4], ids=lambda c: c[0])
def test_dsa_emu_csr_walk_kernel(case, oracle_built, monkeypatch):
    """MAXSUM_LOCAL_SEARCH_GENERIC=1: the CSR-walk kernel (what domains of more than 32 values
    run on) on the instances the slot kernels were just checked on."""
    from emu.build_emu import build
    from oracle.dsa_oracle import OracleDsa
    monkeypatch.setenv("MAXSUM_LOCAL_SEARCH_GENERIC", "1")
    name, make, kw, dsa_kw = case
    compare_dsa(OracleDsa, make(), Params(**kw), dsa_kw, lib_path=build(), steps=(0, 1, 3, 6))


@pytest.mark.parametrize("case", dsa_cases(k=4)[:6], ids=lambda c: c[0])
def test_dsa_emu_slot_kernel_everywhere(case, oracle_built, monkeypatch):
    """MAXSUM_LOCAL_SEARCH_GENERIC=2: the thread-per-variable slot kernel also for the variables the
    packed (lane per constraint) kernel takes by default."""
    from emu.build_emu import build
    from oracle.dsa_oracle import OracleDsa
    monkeypatch.setenv("MAXSUM_LOCAL_SEARCH_GENERIC", "2")
    name, make, kw, dsa_kw = case
    compare_dsa(OracleDsa, make(), Params(**kw), dsa_kw, lib_path=build(), steps=(0, 1, 3, 6))


@pytest.mark.parametrize("case", [c for c in dsa_cases(k=4) if c[0].startswith(("meeting", "mixed_arity3"))], ids=lambda c: c[0])
def test_dsa_emu_strided_slots_without_the_row_view(case, oracle_built, monkeypatch):
    """MAXSUM_LOCAL_SEARCH_ROWS=0: no private row copies (local_search.h, Slots::rows) -- the variables the pack cannot
    take read their D entries per constraint a stride apart, the path of instances whose copies exceed the budget.
    (The default run of the same cases takes the row view: int8 rows for the meeting tables, T rows for the mixed ones.)"""
    from emu.build_emu import build
    from oracle.dsa_oracle import OracleDsa
    monkeypatch.setenv("MAXSUM_LOCAL_SEARCH_ROWS", "0")
    name, make, kw, dsa_kw = case
    compare_dsa(OracleDsa, make(), Params(**kw), dsa_kw, lib_path=build(), steps=(0, 1, 3, 6))
