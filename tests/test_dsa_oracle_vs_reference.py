"""Pins oracle/dsa_oracle.c against the REAL reference: the reference's own DsaComputation objects
(pydcop/algorithms/dsa.py, variants A / B / C, both p_modes) run for exactly n cycles by
oracle/ref_harness.run_reference_dsa, their draws from the unseeded `random` module replaced on
BOTH sides by one counter-based generator (dsa_uniform) -- selected values and held costs, bit for
bit.  Where the reference is on the machine (oracle/stage_reference.locate())."""
import numpy as np
import pytest

from oracle import ref_harness
from pydcop_amd import generators as G
from pydcop_amd.graph import Params

pytestmark = pytest.mark.skipif(not ref_harness.reference_available(), reason="reference tree not present")


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
    ("coloring_soft", lambda: G.random_coloring(40, seed=31), "min"),
    ("coloring_hard", lambda: G.random_coloring(30, seed=32, variant="hard"), "min"),
    ("coloring_max", lambda: G.random_coloring(30, seed=33), "max"),
    ("mixed_arity3", lambda: G.random_mixed(18, 24, seed=34), "min"),
    ("ising_unaries", lambda: G.ising_grid(4, 5, seed=35), "min"),
    ("sparse_isolated", lambda: G.random_coloring(30, avg_degree=1, seed=36), "max"),
    ("unsorted_domains_min", lambda: unsorted_domains_with_ties(G.random_coloring(40, avg_degree=1, seed=37), 37), "min"),
    ("unsorted_domains_max", lambda: unsorted_domains_with_ties(G.random_coloring(40, avg_degree=1, seed=38), 38), "max"),
]


def test_generator_is_the_same_on_both_sides(oracle_built):
    from oracle.dsa_oracle import uniform
    for seed, v, c, k in [(0, 0, 0, 0), (7, 3, 12, 1), (2 ** 40 + 5, 99999, 10 ** 6, 2), (1, 2 ** 30, 5, 1)]:
        assert uniform(seed, v, c, k) == ref_harness.dsa_uniform(seed, v, c, k)
        assert 0.0 <= uniform(seed, v, c, k) < 1.0


@pytest.mark.parametrize("name,make,mode", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("variant,probability,p_mode", [("A", 0.7, "fixed"), ("B", 0.7, "fixed"), ("C", 0.4, "fixed"),
                                                        ("B", 0.7, "arity"), ("C", 1.0, "fixed")])
@pytest.mark.parametrize("cycles", [0, 1, 3, 15])
def test_dsa_oracle_equals_reference(name, make, mode, variant, probability, p_mode, cycles, oracle_built):
    from oracle.dsa_oracle import OracleDsa
    g = make()
    if p_mode == "arity":
        ar = np.diff(g.factor_rowptr)
        edge_factor = np.repeat(np.arange(g.n_factors), ar)
        n_count = np.bincount(g.edge_var, weights=(ar[edge_factor] - 1), minlength=g.n_vars)
        if (n_count == 0).any():
            pytest.skip("a variable without a non-unary constraint divides by zero in the reference's arity mode (dsa.py:260)")
    dcop, _ = ref_harness.flat_to_dcop(g, mode)
    index = {n: i for i, n in enumerate(g.var_names)}
    vals, costs, comps = ref_harness.run_reference_dsa(dcop, cycles, variant, probability, p_mode, seed=11,
                                                       var_index=index)
    o = OracleDsa(g, Params(mode=mode), variant, probability, p_mode, seed=11)
    o.run(cycles)
    idx, cost = o.assignment()
    ref_idx = np.array([g.domains[i].index(vals[n]) for i, n in enumerate(g.var_names)])
    np.testing.assert_array_equal(idx, ref_idx)
    ref_cost = np.array([0.0 if costs[n] is None else costs[n] for n in g.var_names])
    np.testing.assert_array_equal(cost, ref_cost)
