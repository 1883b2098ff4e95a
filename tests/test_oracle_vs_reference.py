"""Pins the C oracle against the REAL reference, run live through
oracle/ref_harness.py.  Where the reference is on the machine
(oracle/stage_reference.locate()); tests/golden/ carries the same pins everywhere, and
tests/test_gpu_vs_reference.py compares the HIP engine with the reference directly on the GPU box."""
import numpy as np
import pytest

from oracle import ref_harness
from pydcop_amd import generators as G
from pydcop_amd.graph import Params
from parity_common import assert_messages_equal_reference

pytestmark = pytest.mark.skipif(not ref_harness.reference_available(),
                                reason="reference tree not present")

def _hard(g, seed, frac, value):
    """Hard constraints as the reference writes them (+-inf table entries, dcop.py:352-365):
    costs_for_factor's mean then produces inf - inf = NaN (maxsum.py:671-674) on both sides."""
    rng = np.random.default_rng(seed)
    t = g.tables.copy()
    t[rng.random(t.shape[0]) < frac] = value
    g.tables = t
    return g

def _hub(seed, nf, n):
    """One variable in nf more binary factors: a degree the reference's scale-free generator produces
    (graphcoloring.py:322-340) -- the serial chains of costs_for_factor (maxsum.py:651-665) at length 3 * nf."""
    from pydcop_amd.graph import FlatGraph
    g = G.random_coloring(n, avg_degree=3, seed=seed)
    rng = np.random.default_rng(seed)
    others = rng.choice(np.arange(1, n), size=nf, replace=False)
    edge_var = np.concatenate([g.edge_var, np.stack([np.zeros(nf, int), others], 1).reshape(-1)])
    rowptr = np.concatenate([g.factor_rowptr, g.factor_rowptr[-1] + 2 * np.arange(1, nf + 1)])
    tables = np.concatenate([g.tables, rng.integers(0, 10, nf * 9).astype(float)])
    toff = np.concatenate([g.table_off, g.table_off[-1] + 9 * np.arange(1, nf + 1)])
    vr, ve = FlatGraph.var_side_from_edges(edge_var, n)
    out = FlatGraph(dom_size=g.dom_size, var_cost=g.var_cost, factor_rowptr=rowptr, edge_var=edge_var,
                    table_off=toff, tables=tables, var_rowptr=vr, var_edges=ve).validate()
    width = len(str(n - 1))
    out.var_names = [f"v{i:0{width}d}" for i in range(n)]
    out.factor_names = [f"c{i:04d}" for i in range(out.n_factors)]
    out.domains = [list(range(int(d))) for d in out.dom_size]
    return out


MAX_T = {"secp_arity5": 7}   # (the reference walks 5 x 5 x 625 assignments per arity-5 factor and cycle in Python)
CASES = [
    ("soft", lambda: G.random_coloring(40, seed=11), "min", {}),
    ("hard", lambda: G.random_coloring(40, seed=12, variant="hard"), "min",
     {"start_messages": "all", "damping_nodes": "vars"}),
    ("mixed", lambda: G.random_mixed(20, 30, seed=13), "max",
     {"start_messages": "leafs_vars", "damping_nodes": "none"}),
    ("meeting", lambda: G.meeting_like(10, dom=4, seed=14), "max",
     {"damping_nodes": "factors", "damping": 0.3, "stability": 0.02}),
    ("hard_inf_nary", lambda: _hard(G.meeting_like(8, n_factors=5, dom=8, seed=15), 15, 0.9, -np.inf), "max", {}),
    ("hard_inf_binary", lambda: _hard(G.random_coloring(30, avg_degree=5, seed=16), 16, 0.3, np.inf), "min",
     {"start_messages": "all"}),
    # round 6: the shapes of the reference's other generators -- a hub of degree 75 (scale-free colourings) and the SECP
    # model (generators.secp_like: D = 5, arity 1..5 with --max_model_size 4)
    ("hub_deg75", lambda: _hub(17, 72, 90), "min", {}),
    ("secp_arity5", lambda: G.secp_like(8, 3, 6, max_model_size=4, seed=26), "min", {"start_messages": "leafs_vars"}),
]


@pytest.mark.parametrize("name,make,mode,params", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("T", [0, 1, 2, 7, 25])
def test_oracle_equals_reference(name, make, mode, params, T, oracle_built):
    if T > MAX_T.get(name, 10 ** 9):
        pytest.skip("bounded for this case")
    g = make()
    dcop, cg = ref_harness.flat_to_dcop(g, mode)
    vals, costs = ref_harness.run_reference_maxsum(dcop, T, params, cg=cg)
    o = oracle_built.OracleMaxSum(g, Params(mode=mode, **params))
    o.run(T)
    idx, belief = o.assignment()
    ref_idx = np.array([g.domains[i].index(vals[n]) for i, n in enumerate(g.var_names)])
    np.testing.assert_array_equal(idx, ref_idx)
    ref_cost = np.array([costs[n] for n in g.var_names], dtype=float)
    np.testing.assert_allclose(belief, ref_cost, rtol=1e-12, atol=1e-12)
    viol, cost = dcop.solution_cost(vals, float("inf"))
    ocost, oviol = o.eval_cost()
    assert oviol == viol and ocost == pytest.approx(cost, rel=1e-12, abs=1e-9)


@pytest.mark.parametrize("name,make,mode,params", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("T", [1, 2, 7, 25])
def test_oracle_messages_and_counters_equal_reference(name, make, mode, params, T, oracle_built):
    """Message-level pin of the synchronous oracle: every receiver's `_costs` and every
    sender's `_prev_messages` (message AND count) of the reference's own computations."""
    if T > MAX_T.get(name, 10 ** 9):
        pytest.skip("bounded for this case")
    g = make()
    dcop, cg = ref_harness.flat_to_dcop(g, mode)
    _, _, comps = ref_harness.run_reference_maxsum(dcop, T, params, cg=cg, return_comps=True)
    ref = ref_harness.reference_message_state(comps, g)
    o = oracle_built.OracleMaxSum(g, Params(mode=mode, **params))
    o.run(T - 1)
    before = o.messages()
    o.run(1)
    assert_messages_equal_reference(ref, o.messages(), before)


def test_reference_unit_pins(oracle_built):
    """tests/unit/test_algorithms_amaxsum.py:77-150 pins factor_costs_for_var on a
    binary factor abs((x1-x2)/2) over domains 0..9 with no received costs:
    costs[5]==0.5? no -- with empty recv the optimum over x2 is 0 for every x1;
    here we pin the same numbers through one engine cycle instead."""
    from pydcop_amd.graph import FlatGraph
    D = 10
    tab = np.abs((np.arange(D)[:, None] - np.arange(D)[None, :]) / 2.0)
    dom = np.array([D, D], dtype=np.int32)
    rowptr, edges = FlatGraph.var_side_from_edges(np.array([0, 1]), 2)
    g = FlatGraph(dom_size=dom, var_cost=np.zeros(2 * D), factor_rowptr=[0, 2],
                  edge_var=[0, 1], table_off=[0, D * D], tables=tab.reshape(-1),
                  var_rowptr=rowptr, var_edges=edges).validate()
    o = oracle_built.OracleMaxSum(g, Params(start_messages="all"))
    _, f2v, _, _ = o.messages()
    np.testing.assert_array_equal(f2v, np.zeros(2 * D))  # min over the other var
    o2 = oracle_built.OracleMaxSum(g, Params(mode="max", start_messages="all"))
    _, f2v, _, _ = o2.messages()
    # max_x2 |x1-x2|/2 : 4.5 at the ends, 2.5 in the middle (x1=5 -> |5-0|/2)
    assert f2v[0] == 4.5 and f2v[5] == 2.5 and f2v[9] == 4.5
