"""Pins oracle/mgm_oracle.c against the REAL reference: the reference's own MgmComputation objects
(pydcop/algorithms/mgm.py) run for exactly R rounds by oracle/ref_harness.run_reference_mgm --
selected values and held costs, bit for bit.  Variable costs sit on a binary grid so that the one
order the reference leaves to PYTHONHASHSEED (its `concerned_vars` set) cannot change a sum.
Where the reference is on the machine (oracle/stage_reference.locate())."""
import numpy as np
import pytest

from oracle import ref_harness
from pydcop_amd import generators as G
from pydcop_amd.graph import Params

pytestmark = pytest.mark.skipif(not ref_harness.reference_available(), reason="reference tree not present")


def grid_costs(g, seed, scale=64.0):
    rng = np.random.default_rng(seed)
    g.var_cost = rng.integers(0, 32, g.var_cost.shape[0]) / scale     # exact sums in any order
    return g


def with_init(g, seed):
    rng = np.random.default_rng(seed)
    g.init_idx = np.array([rng.integers(0, d) if rng.random() < 0.6 else -1 for d in g.dom_size], dtype=np.int32)
    return g



def unsorted_domains_with_ties(g, seed):
    """Domains written in a NON-ascending order (['R', 'G', 'B'], sorted: B < G < R) and own costs on
    two levels, so that the variables without neighbours have cost ties: the reference's
    optimal_cost_value breaks them on the VALUE (min / max over (cost, value) tuples,
    relations.py:1661-1665), not on the position in the domain."""
    rng = np.random.default_rng(seed)
    g.domains = [["R", "G", "B"][:int(d)] for d in g.dom_size]
    g.var_cost = rng.integers(0, 2, g.var_cost.shape[0]) / 64.0
    return g


CASES = [
    ("coloring_soft", lambda: grid_costs(G.random_coloring(40, seed=21), 21), "min"),
    ("coloring_hard_ties", lambda: grid_costs(G.random_coloring(30, seed=22, variant="hard"), 22, 2.0 ** 30), "min"),
    ("coloring_init", lambda: with_init(grid_costs(G.random_coloring(35, seed=23), 23), 23), "min"),
    ("mixed_arity3_max", lambda: grid_costs(G.random_mixed(18, 24, seed=24, float_tables=False), 24), "max"),
    ("mixed_arity3_min", lambda: with_init(grid_costs(G.random_mixed(18, 24, seed=25, float_tables=False), 25), 25), "min"),
    ("ising_unaries", lambda: grid_costs(G.ising_grid(4, 5, seed=26), 26), "min"),
    ("sparse_isolated", lambda: grid_costs(G.random_coloring(30, avg_degree=1, seed=27), 27), "max"),
    ("unsorted_domains_min", lambda: unsorted_domains_with_ties(G.random_coloring(40, avg_degree=1, seed=28), 28), "min"),
    ("unsorted_domains_max", lambda: unsorted_domains_with_ties(G.random_coloring(40, avg_degree=1, seed=29), 29), "max"),
]


@pytest.mark.parametrize("name,make,mode", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("rounds", [0, 1, 2, 5, 12])
def test_mgm_oracle_equals_reference(name, make, mode, rounds, oracle_built):
    from oracle.mgm_oracle import OracleMgm
    g = make()
    if name == "ising_unaries":   # exact table sums too: k on a binary grid
        g.tables = np.round(g.tables * 64) / 64
    dcop, _ = ref_harness.flat_to_dcop(g, mode)
    vals, costs, comps = ref_harness.run_reference_mgm(dcop, rounds)
    o = OracleMgm(g, Params(mode=mode))
    o.run(rounds)
    st = o.state()
    ref_idx = np.array([g.domains[i].index(vals[n]) for i, n in enumerate(g.var_names)])
    np.testing.assert_array_equal(st["idx"], ref_idx)
    for i, n in enumerate(g.var_names):
        if costs[n] is None:
            assert not st["has_cost"][i], n
        else:
            assert st["has_cost"][i] and st["cost"][i] == costs[n], (n, st["cost"][i], costs[n])
    viol, cost = dcop.solution_cost(vals, float("inf"))
    ocost, oviol = o.eval_cost()
    assert oviol == viol and ocost == pytest.approx(cost, rel=1e-12, abs=1e-9)
    assert all(c.cycle_count == rounds + 1 for c in comps.values() if c._neighbors) or rounds == 0
