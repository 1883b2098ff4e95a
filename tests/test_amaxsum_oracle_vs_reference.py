"""Pins oracle/amaxsum_oracle.c against the REAL reference: the reference's own amaxsum
computations (pydcop/algorithms/amaxsum.py) driven first-in-first-out by
oracle/ref_harness.run_reference_amaxsum -- values, costs, the number of messages of every
generation, and every message a computation holds / last sent.  Where the reference is on the
machine (oracle/stage_reference.locate()); tests/golden/amaxsum/*.npz carry the same pins everywhere."""
import numpy as np
import pytest

from oracle import ref_harness
from pydcop_amd import generators as G
from pydcop_amd.graph import Params

pytestmark = pytest.mark.skipif(not ref_harness.reference_available(),
                                reason="reference tree not present")

CASES = [
    ("soft_leafs_vars", lambda: G.random_coloring(30, seed=11), "min", {"start_messages": "leafs_vars"}),
    ("soft_all", lambda: G.random_coloring(30, seed=12), "min", {"start_messages": "all"}),
    ("leafs_only", lambda: G.random_coloring(40, avg_degree=2, seed=13), "min", {}),
    ("hard_vars_damping", lambda: G.random_coloring(25, seed=14, variant="hard"), "min",
     {"start_messages": "all", "damping_nodes": "vars"}),
    ("mixed_max_none", lambda: G.random_mixed(16, 22, seed=15), "max",
     {"start_messages": "leafs_vars", "damping_nodes": "none"}),
    ("meeting_factors", lambda: G.meeting_like(9, dom=4, seed=16), "max",
     {"start_messages": "all", "damping_nodes": "factors", "damping": 0.3, "stability": 0.02}),
    ("ising", lambda: G.ising_grid(4, 5, seed=17), "min", {"start_messages": "leafs_vars"}),
]


def _held(comps, g, kind):
    """The reference's held costs / last sent messages as arrays in the oracle's layout."""
    nm, ne = int(g.msg_off[-1]), g.n_edges
    f_cost, v_cost, f_prev, v_prev = (np.zeros(nm) for _ in range(4))
    f_has, v_has, f_cnt, v_cnt = (np.zeros(ne, dtype=np.uint8) for _ in range(4))
    edge_factor = np.repeat(np.arange(g.n_factors), np.diff(g.factor_rowptr))
    for e in range(ne):
        v, f = int(g.edge_var[e]), int(edge_factor[e])
        vn, fn = g.var_names[v], g.factor_names[f]
        dom = g.domains[v]
        sl = slice(int(g.msg_off[e]), int(g.msg_off[e + 1]))
        fc, vc = comps[fn], comps[vn]
        if vn in fc._costs:
            f_has[e] = 1
            f_cost[sl] = [fc._costs[vn][d] for d in dom]
        if fn in vc._costs:
            v_has[e] = 1
            v_cost[sl] = [vc._costs[fn][d] for d in dom]
        pm, cnt = fc._prev_messages[vn] if vn in fc._prev_messages else (None, 0)
        if pm is not None:
            f_prev[sl] = [pm[d] for d in dom]
        f_cnt[e] = cnt
        pm, cnt = vc._prev_messages[fn] if fn in vc._prev_messages else (None, 0)
        if pm is not None:
            v_prev[sl] = [pm[d] for d in dom]
        v_cnt[e] = cnt
    return dict(f_cost=f_cost, v_cost=v_cost, f_prev=f_prev, v_prev=v_prev,
                f_has=f_has, v_has=v_has, f_cnt=f_cnt, v_cnt=v_cnt)


@pytest.mark.parametrize("name,make,mode,params", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("gens", [0, 1, 2, 5, 12, -1])
def test_amaxsum_oracle_equals_reference(name, make, mode, params, gens, oracle_built):
    from oracle.amaxsum_oracle import OracleAMaxSum
    g = make()
    dcop, cg = ref_harness.flat_to_dcop(g, mode)
    vals, costs, info = ref_harness.run_reference_amaxsum(dcop, gens, params, cg=cg,
                                                          max_messages=None if gens >= 0 else 40_000)
    o = OracleAMaxSum(g, Params(mode=mode, **params))
    n = o.run(gens, -1 if gens >= 0 else 40_000)
    assert n == info["delivered"] and o.pending == info["pending"]
    np.testing.assert_array_equal(o.generation_sizes(), info["generation_sizes"])
    idx, belief = o.assignment()
    ref_idx = np.array([g.domains[i].index(vals[nm]) for i, nm in enumerate(g.var_names)])
    np.testing.assert_array_equal(idx, ref_idx)
    ref_cost = np.array([0.0 if costs[nm] is None else costs[nm] for nm in g.var_names], dtype=float)
    np.testing.assert_array_equal(belief, ref_cost)          # same order of additions: bit for bit
    held, mine = _held(info["comps"], g, None), o.messages()
    for k in held:
        np.testing.assert_array_equal(mine[k], held[k], err_msg=k)
    if gens == -1:
        assert o.pending == 0 or n == 40_000
