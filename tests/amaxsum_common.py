"""Shared amaxsum checks: engine (GPU) or oracle against the golden vectors of the reference."""
import glob
import json
import os

import numpy as np

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
