"""MGM on the GPU (pydcop_amd/csrc/mgm.hip through the mxs_mgm_* C-ABI) against the oracle
(oracle/mgm_oracle.c, pinned against the reference's own MgmComputation): values, held costs,
gains and intended moves bit for bit after 0, 1, 2, 5, 15, 40 rounds, f64 and f32; and a
100k-variable instance."""
import numpy as np
import pytest

from mgm_common import compare_mgm, mgm_cases
from pydcop_amd import generators as G
from pydcop_amd.graph import Params

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", mgm_cases(), ids=lambda c: c[0])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_mgm_bit_exact_vs_oracle(case, dtype, oracle_built):
    from oracle.mgm_oracle import OracleMgm
    name, make, kw = case
    compare_mgm(OracleMgm, make(), Params(dtype=dtype, **kw))


def test_mgm_100k_coloring(oracle_built):
    from oracle.mgm_oracle import OracleMgm
    g = G.random_coloring(100_000, seed=0, names=False)
    compare_mgm(OracleMgm, g, Params(), steps=(1, 20))
    from pydcop_amd.mgm import MgmEngine
    with MgmEngine(g, Params()) as e:
        start = e.eval_cost()[0]
        e.run(60)
        assert e.eval_cost()[0] < 0.6 * start     # MGM is monotone: the cost only goes down


@pytest.mark.parametrize("case", mgm_cases()[:6], ids=lambda c: c[0])
def test_mgm_csr_walk_kernels(case, oracle_built, monkeypatch):
    """MAXSUM_LOCAL_SEARCH_GENERIC=1: the CSR-walk kernels (domains of more than 32 values) on
    the instances of the slot kernels."""
    from oracle.mgm_oracle import OracleMgm
    monkeypatch.setenv("MAXSUM_LOCAL_SEARCH_GENERIC", "1")
    name, make, kw = case
    compare_mgm(OracleMgm, make(), Params(**kw))


@pytest.mark.parametrize("case", mgm_cases()[:6], ids=lambda c: c[0])
def test_mgm_slot_kernels_everywhere(case, oracle_built, monkeypatch):
    """MAXSUM_LOCAL_SEARCH_GENERIC=2: the thread-per-variable slot kernels also for the variables
    the packed (lane per constraint) kernels take by default."""
    from oracle.mgm_oracle import OracleMgm
    monkeypatch.setenv("MAXSUM_LOCAL_SEARCH_GENERIC", "2")
    name, make, kw = case
    compare_mgm(OracleMgm, make(), Params(**kw))


@pytest.mark.parametrize("case", [c for c in mgm_cases() if c[0].startswith(("meeting", "mixed"))], ids=lambda c: c[0])
def test_mgm_strided_slots_without_the_row_view(case, oracle_built, monkeypatch):
    """MAXSUM_LOCAL_SEARCH_ROWS=0: no private row copies (local_search.h, Slots::rows) -- the strided slot reads, the
    path of instances whose copies exceed the budget."""
    
    from oracle.mgm_oracle import OracleMgm
    monkeypatch.setenv("MAXSUM_LOCAL_SEARCH_ROWS", "0")
    name, make, kw = case
    compare_mgm(OracleMgm, make(), Params(**kw), steps=(0, 1, 3, 6))
