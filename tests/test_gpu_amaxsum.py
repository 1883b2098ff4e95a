"""Asynchronous Max-Sum on the GPU (pydcop_amd/csrc/amaxsum.hip through the mxs_amaxsum_* C-ABI)
against the oracle (the reference's amaxsum under FIFO delivery, oracle/amaxsum_oracle.c): every
held / last-sent message, send counter, selection and cost, BIT-EXACT after 0, 1, 2, ... generations
and at quiescence, f64 and f32; and against the golden vectors of the reference itself."""
import os

import pytest

from amaxsum_common import amaxsum_cases, check_golden, compare_amaxsum, golden_files
from pydcop_amd import generators as G
from pydcop_amd.amaxsum import AMaxSumEngine
from pydcop_amd.graph import Params

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", amaxsum_cases(), ids=lambda c: c[0])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_amaxsum_bit_exact_vs_oracle(case, dtype, oracle_built):
    from oracle.amaxsum_oracle import OracleAMaxSum
    name, make, kw = case
    g = make()
    p = Params(dtype=dtype, **kw)
    compare_amaxsum(AMaxSumEngine(g, p), OracleAMaxSum(g, p))


@pytest.mark.parametrize("case", amaxsum_cases()[:7], ids=lambda c: c[0])
def test_amaxsum_per_message_handler(case, oracle_built, monkeypatch):
    """MAXSUM_AMAXSUM_GENERIC=1: every destination on the per-message handler (what large domains,
    n-ary factors and degrees above 64 run on)."""
    from oracle.amaxsum_oracle import OracleAMaxSum
    monkeypatch.setenv("MAXSUM_AMAXSUM_GENERIC", "1")
    name, make, kw = case
    g = make()
    p = Params(**kw)
    compare_amaxsum(AMaxSumEngine(g, p), OracleAMaxSum(g, p), largest=100_000)


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_amaxsum_golden_reference_vectors(path):
    check_golden(lambda g, p: AMaxSumEngine(g, p), path)


def test_amaxsum_errors():
    from pydcop_amd.engine import MaxSumGpuError
    g = G.random_coloring(20, seed=0)
    with pytest.raises(MaxSumGpuError):
        AMaxSumEngine(g, Params(), device=99)


@pytest.mark.parametrize("env", [{"MAXSUM_AMAXSUM_CLEAR_SLOTS": "1"}, {"MAXSUM_AMAXSUM_TWO_SCANS": "1"},
                                 {"MAXSUM_AMAXSUM_ORDER": "dynamic"}, {"MAXSUM_AMAXSUM_ORDER": "static"}, {"MAXSUM_AMAXSUM_CAP_LOOKUP": "1"},
                                 {"MAXSUM_AMAXSUM_CLEAR_SLOTS": "1", "MAXSUM_AMAXSUM_GENERIC": "1"}],
                         ids=lambda e: "+".join(k[15:].lower() + "=" + v for k, v in e.items()))
def test_amaxsum_bookkeeping_variants(env, oracle_built, monkeypatch):
    """The switches of the generation bookkeeping (amaxsum.hip, step / finish): slot words poisoned before the
    handlers run (every handler must write the words of all its output slots -- they are not cleared), the two
    separate scans of very large generations, the destinations re-ordered by queue length / in the static order."""
    from oracle.amaxsum_oracle import OracleAMaxSum
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for name, make, kw in amaxsum_cases(k=2)[:5]:
        g = make()
        p = Params(**kw)
        compare_amaxsum(AMaxSumEngine(g, p), OracleAMaxSum(g, p), first=(1, 2, 3, 6), last_generation=14, largest=20_000)


def test_amaxsum_bench_instance_100k(oracle_built):
    """The instance tools/amaxsum_bench.py and the bench line's amaxsum row run (100k variables, start_messages
    leafs_vars), through the generations of 3, 9, 7 and 14 million messages -- the sizes at which a generation's
    destinations are re-ordered by queue length and the class kernels run thousands of waves: every held / last-sent
    message, counter, selection and cost against the oracle, bit for bit."""
    from amaxsum_common import same_state
    from oracle.amaxsum_oracle import OracleAMaxSum
    g = G.random_coloring(100_000, avg_degree=4, n_colors=3, seed=0, names=False)
    p = Params(start_messages="leafs_vars")
    eng, ora = AMaxSumEngine(g, p), OracleAMaxSum(g, p)
    for gens in (5, 7, 9):
        assert eng.run(gens) == ora.run(gens)
        same_state(eng, ora, f"generations < {gens}")
    assert int(eng.generation_sizes().max()) > 10_000_000
    eng.close(), ora.close()
