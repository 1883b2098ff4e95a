"""MGM: engine (HIP on the GPU, or the emulated build on the CPU) against the oracle, bit for bit."""
import numpy as np

from pydcop_amd import generators as G
from pydcop_amd.graph import Params
from pydcop_amd.mgm import MgmEngine


def with_init(g, seed):
    rng = np.random.default_rng(seed)
    g.init_idx = np.array([rng.integers(0, d) if rng.random() < 0.6 else -1 for d in g.dom_size], dtype=np.int32)
    return g


def shuffled_names(g, seed):
    rng = np.random.default_rng(seed)
    perm = rng.permutation(g.n_vars)
    g.var_names = [f"n{int(perm[i]):05d}" for i in range(g.n_vars)]   # lexic ties follow the NAMES
    return g


def unsorted_domains_with_ties(g, seed):
    """Domains written in a non-ascending order and own costs on two levels: the variables without
    neighbours start on cost ties, broken on the VALUE (FlatGraph.value_rank, relations.py:1661-1665)."""
    rng = np.random.default_rng(seed)
    g.domains = [["R", "G", "B"][:int(d)] for d in g.dom_size]
    g.var_cost = rng.integers(0, 2, g.var_cost.shape[0]) / 64.0
    assert g.value_rank() is not None
    return g


def repeated_pairs_and_unaries(n_vars, seed):
    """Integer tables (the int8 records of the packed view, whose lanes MGM lays out in the concerned
    list's order), with what that order has to cope with: several constraints between the SAME pair of
    variables (both orientations), unary constraints, variables below and above all their neighbours."""
    from pydcop_amd.generators import _finish
    rng = np.random.default_rng(seed)
    g = G.random_coloring(n_vars, avg_degree=3, seed=seed)
    D = 3
    pairs = g.edge_var.reshape(-1, 2)
    again = pairs[rng.random(pairs.shape[0]) < 0.35][:, ::-1]           # the same pair once more, flipped
    thrice = again[rng.random(again.shape[0]) < 0.3][:, ::-1]
    un = rng.choice(n_vars, size=n_vars // 3, replace=False)
    scopes = [list(p) for p in pairs] + [list(p) for p in again] + [list(p) for p in thrice] + [[int(v)] for v in un]
    order = rng.permutation(len(scopes))                                 # unary constraints anywhere in a variable's list
    scopes = [scopes[i] for i in order]
    tabs = [rng.integers(-5, 10, size=D ** len(sc)).astype(np.float64) for sc in scopes]
    rowptr = np.zeros(len(scopes) + 1, dtype=np.int32)
    np.cumsum([len(sc) for sc in scopes], out=rowptr[1:])
    toff = np.zeros(len(scopes) + 1, dtype=np.int64)
    np.cumsum([t.size for t in tabs], out=toff[1:])
    return _finish(g.dom_size, g.var_cost, rowptr, np.array([v for sc in scopes for v in sc], dtype=np.int32),
                   np.concatenate(tabs), toff)


def mgm_cases():
    return [
        ("coloring_soft", lambda: G.random_coloring(400, seed=21), {}),
        ("coloring_hard_ties", lambda: shuffled_names(G.random_coloring(300, seed=22, variant="hard"), 22), {}),
        ("coloring_init_max", lambda: with_init(G.random_coloring(350, seed=23), 23), {"mode": "max"}),
        ("mixed_arity3", lambda: G.random_mixed(120, 200, seed=24), {}),
        ("mixed_arity3_max", lambda: with_init(G.random_mixed(80, 120, seed=25, float_tables=False), 25), {"mode": "max"}),
        ("ising_unaries", lambda: G.ising_grid(12, 10, seed=26), {}),
        ("sparse_isolated", lambda: G.random_coloring(300, avg_degree=1, seed=27), {"mode": "max"}),
        ("unsorted_domains", lambda: unsorted_domains_with_ties(G.random_coloring(200, avg_degree=1, seed=32), 32), {}),
        ("repeated_pairs_unaries", lambda: repeated_pairs_and_unaries(300, 34), {}),
        ("repeated_pairs_unaries_max", lambda: with_init(repeated_pairs_and_unaries(200, 35), 35), {"mode": "max"}),
        ("unsorted_domains_max", lambda: unsorted_domains_with_ties(G.random_coloring(200, avg_degree=1, seed=33), 33), {"mode": "max"}),
        ("meeting_d6", lambda: G.meeting_like(40, dom=6, seed=28), {"mode": "max"}),
        # the wider register arrays of the slot kernels (16, 32 values) and the CSR-walk kernel beyond
        ("meeting_d12", lambda: G.meeting_like(24, dom=12, seed=29), {"mode": "max"}),
        ("meeting_d24", lambda: with_init(G.meeting_like(18, dom=24, seed=30), 30), {}),
        ("meeting_d35", lambda: G.meeting_like(12, dom=35, seed=31), {"mode": "max"}),
    ]


def compare_mgm(oracle_cls, graph, params, lib_path=None, steps=(0, 1, 1, 3, 10, 25)):
    eng = MgmEngine(graph, params, lib_path=lib_path)
    ora = oracle_cls(graph, params)
    done = 0
    for n in steps:
        eng.run(n), ora.run(n)
        done += n
        assert eng.cycle_count == ora.cycle_count == done
        se, so = eng.state(), ora.state()
        for k in ("idx", "has_cost", "cost", "gain", "new"):
            np.testing.assert_array_equal(se[k], so[k], err_msg=f"{k} after {done} rounds")
        ce, co = eng.eval_cost(), ora.eval_cost()
        assert ce[1] == co[1] and abs(ce[0] - co[0]) <= 1e-9 * max(1.0, abs(co[0]))
    eng.reset(), ora.reset()
    eng.run(4), ora.run(4)
    np.testing.assert_array_equal(eng.state()["idx"], ora.state()["idx"])
    np.testing.assert_array_equal(eng.state()["cost"], ora.state()["cost"])
    eng.close(), ora.close()
