"""DSA on the GPU (pydcop_amd/csrc/dsa.hip through the mxs_dsa_* C-ABI) against the oracle
(oracle/dsa_oracle.c, pinned against the reference's own DsaComputation under the keyed generator):
values and held costs bit for bit after 0, 1, 2, 5, 15, 40 cycles, variants A / B / C, f64 and f32;
and a 100k-variable instance."""
import pytest

from dsa_common import compare_dsa, dsa_cases
from pydcop_amd import generators as G
from pydcop_amd.graph import Params

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", dsa_cases(), ids=lambda c: c[0])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_dsa_bit_exact_vs_oracle(case, dtype, oracle_built):
    from oracle.dsa_oracle import OracleDsa
    name, make, kw, dsa_kw = case
    compare_dsa(OracleDsa, make(), Params(dtype=dtype, **kw), dsa_kw)


def test_dsa_100k_coloring(oracle_built):
    from oracle.dsa_oracle import OracleDsa
    from pydcop_amd.dsa import DsaEngine
    g = G.random_coloring(100_000, seed=0, names=False)
    compare_dsa(OracleDsa, g, Params(), dict(variant="B", probability=0.7), steps=(1, 20))
    with DsaEngine(g, Params(), seed=3) as e:
        start = e.eval_cost()[0]
        e.run(60)
        assert e.eval_cost()[0] < 0.6 * start


@pytest.mark.parametrize("case", dsa_cases()[:5], ids=lambda c: c[0])
def test_dsa_csr_walk_kernel(case, oracle_built, monkeypatch):
    """MAXSUM_LOCAL_SEARCH_GENERIC=1: the CSR-walk kernel (domains of more than 32 values) on
    the instances of the slot kernels."""
    from oracle.dsa_oracle import OracleDsa
    monkeypatch.setenv("MAXSUM_LOCAL_SEARCH_GENERIC", "1")
    name, make, kw, dsa_kw = case
    compare_dsa(OracleDsa, make(), Params(**kw), dsa_kw)


@pytest.mark.parametrize("case", dsa_cases()[:6], ids=lambda c: c[0])
def test_dsa_slot_kernel_everywhere(case, oracle_built, monkeypatch):
    """MAXSUM_LOCAL_SEARCH_GENERIC=2: the thread-per-variable slot kernel also for the variables the
    packed (lane per constraint) kernel takes by default."""
    from oracle.dsa_oracle import OracleDsa
    monkeypatch.setenv("MAXSUM_LOCAL_SEARCH_GENERIC", "2")
    name, make, kw, dsa_kw = case
    compare_dsa(OracleDsa, make(), Params(**kw), dsa_kw)


@pytest.mark.parametrize("case", [c for c in dsa_cases() if c[0].startswith(("meeting", "mixed_arity3"))], ids=lambda c: c[0])
def test_dsa_strided_slots_without_the_row_view(case, oracle_built, monkeypatch):
    """MAXSUM_LOCAL_SEARCH_ROWS=0: no private row copies (local_search.h, Slots::rows) -- the variables the pack cannot
    take read their D entries per constraint a stride apart, the path of instances whose copies exceed the budget.
    (The default run of the same cases takes the row view: int8 rows for the meeting tables, T rows for the mixed ones.)"""
    
    from oracle.dsa_oracle import OracleDsa
    monkeypatch.setenv("MAXSUM_LOCAL_SEARCH_ROWS", "0")
    name, make, kw, dsa_kw = case
    compare_dsa(OracleDsa, make(), Params(**kw), dsa_kw, steps=(0, 1, 3, 6))
