"""Asynchronous Max-Sum on the GPU (pydcop_amd/csrc/amaxsum.hip through the mxs_amaxsum_* C-ABI)
against the oracle (the reference's amaxsum under FIFO delivery, oracle/amaxsum_oracle.c): every
held / last-sent message, send counter, selection and cost, BIT-EXACT after 0, 1, 2, ... generations
and at quiescence, f64 and f32; and against the golden vectors of the reference itself."""
import os

import numpy as np
import pytest

from amaxsum_common import check_golden, golden_files
from pydcop_amd import generators as G
from pydcop_amd.amaxsum import AMaxSumEngine
from pydcop_amd.graph import Params

pytestmark = pytest.mark.gpu


def with_init(g, seed):
    rng = np.random.default_rng(seed)
    g.init_idx = np.where(rng.random(g.n_vars) < 0.3, rng.integers(0, 2, g.n_vars), -1).astype(np.int32)
    return g


CASES = [
    ("coloring_leafs_vars", lambda: G.random_coloring(200, seed=1), {"start_messages": "leafs_vars"}),
    ("coloring_all", lambda: G.random_coloring(150, seed=2), {"start_messages": "all"}),
    ("coloring_leafs_only", lambda: G.random_coloring(300, avg_degree=2, seed=3), {}),
    ("hard_vars_damping", lambda: G.random_coloring(80, seed=4, variant="hard"),
     {"start_messages": "all", "damping_nodes": "vars"}),
    ("mixed_max_none", lambda: G.random_mixed(60, 90, seed=5),
     {"mode": "max", "start_messages": "leafs_vars", "damping_nodes": "none"}),
    ("meeting_arity3", lambda: G.meeting_like(20, dom=5, seed=6),
     {"mode": "max", "start_messages": "all", "damping_nodes": "factors", "damping": 0.3, "stability": 0.02}),
    ("ising", lambda: G.ising_grid(8, 9, seed=7), {"start_messages": "leafs_vars"}),
    ("init_values", lambda: with_init(G.random_coloring(100, n_colors=2, seed=8), 8), {"start_messages": "all"}),
    ("deg12", lambda: G.random_coloring(60, avg_degree=12, seed=9), {"start_messages": "leafs_vars", "stability": 0.3}),
]


def _same(eng, ora, what):
    me, mo = eng.messages(), ora.messages()
    for k in mo:
        np.testing.assert_array_equal(me[k], mo[k], err_msg=f"{k} {what}")
    np.testing.assert_array_equal(eng.assignment()[0], ora.assignment()[0], err_msg=what)
    np.testing.assert_array_equal(eng.assignment()[1], ora.assignment()[1], err_msg=what)
    np.testing.assert_array_equal(eng.generation_sizes(), ora.generation_sizes(), err_msg=what)
    assert eng.pending == ora.pending and eng.delivered == ora.delivered, what
    ce, co = eng.eval_cost(), ora.eval_cost()
    assert ce[1] == co[1] and abs(ce[0] - co[0]) <= 1e-9 * max(1.0, abs(co[0]))


@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_amaxsum_bit_exact_vs_oracle(case, dtype, oracle_built):
    from oracle.amaxsum_oracle import OracleAMaxSum
    name, make, kw = case
    g = make()
    p = Params(dtype=dtype, **kw)
    eng, ora = AMaxSumEngine(g, p), OracleAMaxSum(g, p)
    _same(eng, ora, "after start")
    for gens in (1, 2, 3, 6, 12):          # run(G): generations 0 .. G-1 delivered
        assert eng.run(gens) == ora.run(gens)
        _same(eng, ora, f"generations < {gens}")
    # on to quiescence -- or generation 80, or (instances whose message count explodes: hard
    # tables, high degrees) until a generation exceeds 300k messages
    gens = 12
    while ora.pending and gens < 80 and ora.pending <= 300_000:
        gens = min(80, gens + 8)
        for g1 in range(gens - 8, gens):
            if ora.pending > 300_000:
                break
            assert eng.run(g1 + 1) == ora.run(g1 + 1)
    _same(eng, ora, "at the end")
    eng.reset(), ora.reset()
    assert eng.run(4) == ora.run(4)
    _same(eng, ora, "after reset")
    eng.close(), ora.close()


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_amaxsum_golden_reference_vectors(path):
    check_golden(lambda g, p: AMaxSumEngine(g, p), path)


def test_amaxsum_errors():
    from pydcop_amd.engine import MaxSumGpuError
    g = G.random_coloring(20, seed=0)
    with pytest.raises(MaxSumGpuError):
        AMaxSumEngine(g, Params(), device=99)
