"""Shared amaxsum checks: engine (GPU, or the emulated build on the CPU) against the oracle bit for
bit, and engine or oracle against the golden vectors of the reference."""
import glob
import json
import os

import numpy as np

from pydcop_amd import generators as G
from pydcop_amd.graph import FlatGraph, Params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "amaxsum")


def golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "*.npz")))


def load(path):
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    g = FlatGraph(dom_size=z["dom_size"], var_cost=z["var_cost"], factor_rowptr=z["factor_rowptr"],
                  edge_var=z["edge_var"], table_off=z["table_off"], tables=z["tables"],
                  var_rowptr=z["var_rowptr"], var_edges=z["var_edges"],
                  init_idx=z["init_idx"] if "init_idx" in z.files else None).validate()
    return g, meta, z["ref_idx"], z["ref_cost"]


def check_golden(make_engine, path):
    """`make_engine(graph, Params)` -> object with run / assignment / generation_sizes / pending /
    eval_cost: after the fixture's number of generations it holds what the reference held."""
    g, meta, ref_idx, ref_cost = load(path)
    eng = make_engine(g, Params(mode=meta["mode"], **meta["params"]))
    n = eng.run(meta["generations"])
    assert n == meta["delivered"] and eng.pending == meta["pending"]
    np.testing.assert_array_equal(eng.generation_sizes(), meta["generation_sizes"])
    idx, belief = eng.assignment()
    np.testing.assert_array_equal(idx, ref_idx)
    np.testing.assert_allclose(belief, ref_cost, rtol=1e-5, atol=1e-5)   # the north-star tolerance
    cost, viol = eng.eval_cost()
    assert viol == meta["violation"] and abs(cost - meta["cost"]) <= 1e-5 * max(1.0, abs(meta["cost"]))
    eng.close()


def with_init(g, seed):
    rng = np.random.default_rng(seed)
    g.init_idx = np.where(rng.random(g.n_vars) < 0.3, rng.integers(0, 2, g.n_vars), -1).astype(np.int32)
    return g


def star_coloring(n_leaves, n_extra, seed, n_colors=3):
    """A hub (variable 0) with `n_leaves` binary constraints + a few random ones among the leaves:
    degrees above 64 leave the wave-per-variable kernels for the per-message handler."""
    from pydcop_amd.generators import _finish
    rng = np.random.default_rng(seed)
    n_vars = n_leaves + 1
    pairs = [(0, i) if rng.random() < 0.5 else (i, 0) for i in range(1, n_vars)]
    seen = set()
    while len(seen) < n_extra:
        a, b = (int(x) for x in rng.integers(1, n_vars, 2))
        if a != b and (min(a, b), max(a, b)) not in seen:
            seen.add((min(a, b), max(a, b)))
    pairs += sorted(seen)
    nf, D = len(pairs), n_colors
    tables = rng.integers(0, 10, size=(nf, D, D)).astype(np.float64)
    return _finish(np.full(n_vars, D, dtype=np.int32), rng.uniform(0.0, 0.01, size=n_vars * D),
                   np.arange(0, 2 * nf + 1, 2, dtype=np.int32), np.array(pairs, dtype=np.int32).reshape(-1),
                   tables.reshape(-1), np.arange(0, (nf + 1) * D * D, D * D, dtype=np.int64))


def amaxsum_cases(k=1):
    """k > 1: graphs k times smaller (the emulated engine of the CPU tests is slow)."""
    return [
        ("coloring_leafs_vars", lambda: G.random_coloring(200 // k, seed=1), {"start_messages": "leafs_vars"}),
        ("coloring_all", lambda: G.random_coloring(150 // k, seed=2), {"start_messages": "all"}),
        ("coloring_leafs_only", lambda: G.random_coloring(300 // k, avg_degree=2, seed=3), {}),
        ("hard_vars_damping", lambda: G.random_coloring(80 // k, seed=4, variant="hard"),
         {"start_messages": "all", "damping_nodes": "vars"}),
        ("mixed_max_none", lambda: G.random_mixed(60 // k, 90 // k, seed=5),
         {"mode": "max", "start_messages": "leafs_vars", "damping_nodes": "none"}),
        ("meeting_arity3", lambda: G.meeting_like(20, dom=5, seed=6),
         {"mode": "max", "start_messages": "all", "damping_nodes": "factors", "damping": 0.3, "stability": 0.02}),
        ("ising", lambda: G.ising_grid(8, 9, seed=7), {"start_messages": "leafs_vars"}),
        ("init_values", lambda: with_init(G.random_coloring(100 // k, n_colors=2, seed=8), 8), {"start_messages": "all"}),
        ("deg12", lambda: G.random_coloring(60, avg_degree=12 if k == 1 else 9, seed=9),
         {"start_messages": "leafs_vars", "stability": 0.3}),
        ("d4_deg20_hub", lambda: G.random_coloring(40, avg_degree=20 if k == 1 else 14, n_colors=4, seed=10),
         {"start_messages": "leafs_vars", "stability": 0.5}),
        ("hub_deg70", lambda: star_coloring(70, 12, seed=11), {"start_messages": "leafs_vars", "stability": 0.3}),
    ]


def same_state(eng, ora, what):
    me, mo = eng.messages(), ora.messages()
    for k in mo:
        np.testing.assert_array_equal(me[k], mo[k], err_msg=f"{k} {what}")
    np.testing.assert_array_equal(eng.assignment()[0], ora.assignment()[0], err_msg=what)
    np.testing.assert_array_equal(eng.assignment()[1], ora.assignment()[1], err_msg=what)
    np.testing.assert_array_equal(eng.generation_sizes(), ora.generation_sizes(), err_msg=what)
    assert eng.pending == ora.pending and eng.delivered == ora.delivered, what
    ce, co = eng.eval_cost(), ora.eval_cost()
    assert ce[1] == co[1] and abs(ce[0] - co[0]) <= 1e-9 * max(1.0, abs(co[0]))


def compare_amaxsum(eng, ora, first=(1, 2, 3, 6, 12), last_generation=80, largest=300_000):
    """Every held / last-sent message, counter, selection and cost after 0, 1, 2, ... generations
    (run(G): generations 0 .. G-1 delivered), then on to quiescence -- or `last_generation`, or
    (instances whose message count explodes: hard tables, high degrees) until a generation
    exceeds `largest` messages -- and after a reset."""
    same_state(eng, ora, "after start")
    for gens in first:
        if ora.pending > largest:
            break
        assert eng.run(gens) == ora.run(gens)
        same_state(eng, ora, f"generations < {gens}")
    gens = ora.generation
    while ora.pending and gens < last_generation and ora.pending <= largest:
        gens += 1
        assert eng.run(gens) == ora.run(gens)
    same_state(eng, ora, "at the end")
    eng.reset(), ora.reset()
    assert eng.run(4) == ora.run(4)
    same_state(eng, ora, "after reset")
    eng.close(), ora.close()
