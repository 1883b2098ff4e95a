"""Does a locality ordering of the variables help the gathers?  
The engine keeps the caller's order inside a degree class (stable sort), so relabelling the
variables of an instance along a reverse Cuthill-McKee / breadth-first sweep before handing it over
orders every class by that sweep; factors follow their first scope variable.  Times the sweep on the
original and on the relabelled instance (same instance up to names: same cost after the same
number of cycles).   usage: python tools/locality_ab.py [workload] [dtype ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from scipy.sparse import coo_matrix  # noqa: E402
from scipy.sparse.csgraph import breadth_first_order, reverse_cuthill_mckee  # noqa: E402

from bench import make_workload  # noqa: E402
from pydcop_amd.engine import MaxSumEngine  # noqa: E402
from pydcop_amd.graph import FlatGraph, Params  # noqa: E402


def relabel(g, order):
    """order[new] = old variable."""
    new_of = np.empty(g.n_vars, dtype=np.int64)
    new_of[order] = np.arange(g.n_vars)
    D = g.dom_size
    assert (D == D[0]).all(), "uniform domains only (benchmark instances)"
    d = int(D[0])
    var_cost = g.var_cost.reshape(g.n_vars, d)[order].reshape(-1)
    edge_var = new_of[g.edge_var].astype(np.int32)
    rowptr, edges = FlatGraph.var_side_from_edges(edge_var, g.n_vars)
    return FlatGraph(dom_size=D.copy(), var_cost=var_cost, factor_rowptr=g.factor_rowptr, edge_var=edge_var,
                     table_off=g.table_off, tables=g.tables, var_rowptr=rowptr, var_edges=edges).validate()


def adjacency(g):
    ar = np.diff(g.factor_rowptr)
    f = np.flatnonzero(ar == 2)
    a, b = g.edge_var[g.factor_rowptr[f]], g.edge_var[g.factor_rowptr[f] + 1]
    n = g.n_vars
    return coo_matrix((np.ones(2 * len(f), dtype=np.int8), (np.concatenate([a, b]), np.concatenate([b, a]))), shape=(n, n)).tocsr()


def time_it(g, mode, dtype, steps):
    with MaxSumEngine(g, Params(mode=mode, dtype=dtype)) as eng:
        eng.run(steps // 5)
        eng.sync()
        ms = eng.run_timed(steps)
        eng.sync()
        cost = eng.eval_cost()[0]
    return 1e3 * ms / steps, cost


if __name__ == "__main__":
    workload = sys.argv[1] if len(sys.argv) > 1 else "coloring_1m_deg6"
    dtypes = sys.argv[2:] or ["f32", "f64"]
    g, mode = make_workload(workload)
    A = adjacency(g)
    t0 = time.perf_counter()
    orders = {"rcm": np.asarray(reverse_cuthill_mckee(A, symmetric_mode=True), dtype=np.int64)}
    bfs = breadth_first_order(A, 0, directed=False, return_predecessors=False)
    rest = np.setdiff1d(np.arange(g.n_vars), bfs)
    orders["bfs"] = np.concatenate([bfs, rest]).astype(np.int64)
    t1 = time.perf_counter()
    steps = 300 if g.n_vars >= 500_000 else 2000
    for dtype in dtypes:
        us0, c0 = time_it(g, mode, dtype, steps)
        rec = {"workload": workload, "dtype": dtype, "steps": steps, "original_us": round(us0, 2), "ordering_wall_s": round(t1 - t0, 1)}
        for name, order in orders.items():
            us, c = time_it(relabel(g, order), mode, dtype, steps)
            rec[name + "_us"] = round(us, 2)
            rec[name + "_same_cost"] = bool(abs(c - c0) <= 1e-6 * max(1.0, abs(c0)))
        print(json.dumps(rec), flush=True)
