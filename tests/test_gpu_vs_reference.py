"""The HIP engine against the REFERENCE ITSELF, live, with no oracle in between: the reference's
own MaxSum{Factor,Variable}Computation objects (pydcop/algorithms/maxsum.py) run for exactly T
cycles by oracle/ref_harness.py on the GPU box's host (the reference travels as the git-ignored
archive oracle/_ref/, packed by build()), the same instance for T cycles on the MI355X through
the C-ABI; compared: every message a computation last sent (`_prev_messages`, message AND count),
every message it holds (`_costs`), the selected values (identical), their costs (1e-12: the
reference sums select_value in dict-arrival order, maxsum.py:609) and DCOP.solution_cost."""
import numpy as np
import pytest

from oracle import ref_harness
from parity_common import assert_messages_equal_reference
from pydcop_amd import generators as G
from pydcop_amd.engine import MaxSumEngine
from pydcop_amd.graph import Params

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref_harness.reference_available(),
                                 reason="no reference: oracle/_ref/ was not staged by build()")]

def _hard(g, seed, frac, value):
    """Hard constraints as the reference writes them (+-inf table entries, dcop.py:352-365):
    costs_for_factor's mean then produces inf - inf = NaN (maxsum.py:671-674) on both sides."""
    rng = np.random.default_rng(seed)
    t = g.tables.copy()
    t[rng.random(t.shape[0]) < frac] = value
    g.tables = t
    return g

import test_oracle_vs_reference as _ovr  # noqa: E402  (the hub instance builder)

CASES = [
    ("soft", lambda: G.random_coloring(60, seed=21), "min", {}),
    ("hard_all_vars", lambda: G.random_coloring(40, seed=22, variant="hard"), "min",
     {"start_messages": "all", "damping_nodes": "vars"}),
    ("mixed_max", lambda: G.random_mixed(20, 30, seed=23), "max",
     {"start_messages": "leafs_vars", "damping_nodes": "none"}),
    ("meeting_nary", lambda: G.meeting_like(10, dom=4, seed=24), "max",
     {"damping_nodes": "factors", "damping": 0.3, "stability": 0.02}),
    ("meeting_d24_wide", lambda: G.meeting_like(6, dom=24, seed=25), "max", {}),
    ("ising", lambda: G.ising_grid(6, 5, seed=26), "min", {}),
    ("deg6_d4", lambda: G.random_coloring(50, avg_degree=6, n_colors=4, seed=27), "min", {"damping_nodes": "factors"}),
    # +-inf tables -> NaN messages, through the workgroup-per-factor and wide-variable kernels
    ("hard_inf_nary_d8", lambda: _hard(G.meeting_like(8, n_factors=5, dom=8, seed=28), 28, 0.9, -np.inf), "max", {}),
    ("hard_inf_wide_deg30", lambda: _hard(G.random_coloring(40, avg_degree=30, n_colors=6, seed=29), 29, 0.5, np.inf),
     "min", {"start_messages": "all"}),
    # round 6: the hub class (a degree the reference's scale-free generator produces) and the small-domain lane-group kernel
    # (the reference's SECP model with arity 3 / 4 / 5 constraints) against the reference's own computations
    ("hub_deg75", lambda: _ovr._hub(17, 72, 90), "min", {}),
    ("hub_deg140_max", lambda: _ovr._hub(31, 140, 160), "max", {"start_messages": "all"}),
    ("secp_arity5", lambda: G.secp_like(8, 3, 6, max_model_size=4, seed=26), "min", {"start_messages": "leafs_vars"}),
]


@pytest.mark.parametrize("name,make,mode,params", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("T", [1, 3, 12])
def test_hip_engine_equals_the_reference(name, make, mode, params, T):
    if name == "secp_arity5" and T > 3:
        pytest.skip("the reference walks 15 625 assignments per arity-5 factor and cycle in Python")
    g = make()
    dcop, cg = ref_harness.flat_to_dcop(g, mode)
    vals, costs, comps = ref_harness.run_reference_maxsum(dcop, T, params, cg=cg, return_comps=True)
    ref = ref_harness.reference_message_state(comps, g)
    with MaxSumEngine(g, Params(mode=mode, **params)) as eng:
        eng.run(T - 1)
        before = eng.messages()
        eng.run(1)
        assert_messages_equal_reference(ref, eng.messages(), before)
        idx, belief = eng.assignment()
        ref_idx = np.array([g.domains[i].index(vals[n]) for i, n in enumerate(g.var_names)])
        np.testing.assert_array_equal(idx, ref_idx)
        np.testing.assert_allclose(belief, np.array([costs[n] for n in g.var_names], dtype=float),
                                   rtol=1e-12, atol=1e-12)
        viol, cost = dcop.solution_cost(vals, float("inf"))
        ecost, eviol = eng.eval_cost()
        assert eviol == viol and ecost == pytest.approx(cost, rel=1e-12, abs=1e-9)


def test_dynamic_changes_on_the_hip_engine_equal_the_reference(oracle_built):
    """SURVEY 8(f).3 on hardware: the reference's own change_factor_function (same scope, permuted
    dimension order) and relation.slice against mxs_update_factor_table / mxs_slice_factor of the
    HIP library -- the cases of tests/test_dynamic_vs_reference.py with lib_path=None."""
    import test_dynamic_vs_reference as D
    for _, make, mode, params in D.CASES:
        D.check_change_factor_function(make, mode, params, oracle_built, None)
    D.check_external_slice(oracle_built, None)
    for _, make, mode, params in D.ACASES:   # the method where the reference defines it: asynchronous Max-Sum
        D.check_reference_dynamic_class_under_amaxsum(make, mode, params, None)
