"""Shared checks of pydcop_amd.dynamic (engine on the GPU, or the emulated build on the CPU)
against the oracle, plus a pure-Python restatement of the scope-change semantics of
pydcop/algorithms/maxsum_dynamic.py:234-271, 352-405 on dict-of-dict messages (TEST
INFRASTRUCTURE: the reference's own classes are marked broken, maxsum_dynamic.py:60)."""
import numpy as np

from pydcop_amd import generators as G
from pydcop_amd.dynamic import DynamicMaxSum, rescope_factor
from pydcop_amd.graph import Params


def reference_style_rescope(graph, state, f, new_scope, new_table, mode):
    """What the reference's handlers do, message by message, with its own data structures:
    factor._costs / _prev_messages, variable._factors / _costs / _prev_messages."""
    g = graph
    off = g.msg_off
    e0, e1 = int(g.factor_rowptr[f]), int(g.factor_rowptr[f + 1])
    old_scope = [int(x) for x in g.edge_var[e0:e1]]
    # the factor's view
    costs = {v: list(state["v2f"][off[e0 + i]:off[e0 + i + 1]]) for i, v in enumerate(old_scope)}
    prev = {v: (list(state["f2v"][off[e0 + i]:off[e0 + i + 1]]), int(state["count_f2v"][e0 + i]))
            for i, v in enumerate(old_scope)}
    removed = [v for v in old_scope if v not in new_scope]
    added = [v for v in new_scope if v not in old_scope]
    for v in removed:                       # maxsum_dynamic.py:251-255
        costs.pop(v, None)
        prev.pop(v, None)
    for v in added:                         # :256-257
        costs[v] = [0.0] * int(g.dom_size[v])
    table = np.asarray(new_table, dtype=np.float64).reshape([int(g.dom_size[v]) for v in new_scope])
    add_msgs = {}
    for v in added:                         # :290-313 _send_add_var_msg -> factor_costs_for_var
        pos = new_scope.index(v)
        out = []
        for d in range(int(g.dom_size[v])):
            best = None
            others = [u for u in new_scope if u != v]
            for assign in np.ndindex(*[int(g.dom_size[u]) for u in others]):
                full = [0] * len(new_scope)
                full[pos] = d
                sum_cost = 0.0
                for u, a in zip(others, assign):
                    full[new_scope.index(u)] = a
                    sum_cost += costs[u][a]
                val = float(table[tuple(full)]) + sum_cost
                if best is None or (val < best if mode == "min" else val > best):
                    best = val
            out.append(best)
        add_msgs[v] = out
    return {"removed": removed, "added": added, "factor_costs": costs, "factor_prev": prev, "add_msgs": add_msgs}


def check_rescope_semantics(seed=0):
    """rescope_factor against the dict-based restatement, field by field."""
    g = G.random_mixed(14, 16, seed=seed, max_arity=3, dom_choices=(2, 3, 4), names=False)
    rng = np.random.default_rng(seed)
    nm, ne = int(g.msg_off[-1]), g.n_edges
    state = {"v2f": rng.uniform(-3, 3, nm), "f2v": rng.uniform(-3, 3, nm),
             "count_v2f": rng.integers(0, 5, ne).astype(np.uint8), "count_f2v": rng.integers(0, 5, ne).astype(np.uint8),
             "idx": np.zeros(g.n_vars, dtype=np.int32), "belief": np.zeros(g.n_vars), "cycles": 9}
    f = int(np.flatnonzero(np.diff(g.factor_rowptr) >= 2)[0])
    e0, e1 = int(g.factor_rowptr[f]), int(g.factor_rowptr[f + 1])
    old = [int(x) for x in g.edge_var[e0:e1]]
    fresh = [v for v in range(g.n_vars) if v not in old][:2]
    new_scope = [fresh[0], old[-1], fresh[1]]          # drops old[:-1], keeps one, adds two
    table = rng.uniform(-4, 4, int(np.prod(g.dom_size[new_scope])))
    for mode in ("min", "max"):
        ng, ns = rescope_factor(g, state, f, new_scope, table, mode)
        ref = reference_style_rescope(g, state, f, new_scope, table, mode)
        no = ng.msg_off
        ne0 = int(ng.factor_rowptr[f])
        assert [int(x) for x in ng.edge_var[ne0:int(ng.factor_rowptr[f + 1])]] == new_scope
        for i, v in enumerate(new_scope):
            e = ne0 + i
            np.testing.assert_array_equal(ns["v2f"][no[e]:no[e + 1]], ref["factor_costs"][v])
            if v in ref["added"]:
                np.testing.assert_array_equal(ns["f2v"][no[e]:no[e + 1]], ref["add_msgs"][v])
                assert ns["count_f2v"][e] == 0 and ns["count_v2f"][e] == 0
                assert int(ng.var_edges[ng.var_rowptr[v + 1] - 1]) == e      # appended to the variable's list
            else:
                np.testing.assert_array_equal(ns["f2v"][no[e]:no[e + 1]], ref["factor_prev"][v][0])
                assert ns["count_f2v"][e] == ref["factor_prev"][v][1]
        for v in ref["removed"]:                       # REMOVE: factor gone, previous messages cleared
            ks = ng.var_edges[ng.var_rowptr[v]:ng.var_rowptr[v + 1]]
            assert all(int(ng.edge_var[k]) == v for k in ks) and not any(ne0 <= k < ne0 + len(new_scope) for k in ks)
            assert all(ns["count_v2f"][k] == 0 for k in ks)
        # every other edge keeps messages and counters
        untouched = [e for e in range(g.n_edges) if not (e0 <= e < e1) and int(g.edge_var[e]) not in ref["removed"]]
        shift = len(new_scope) - len(old)
        for e in untouched:
            e2 = e if e < e0 else e + shift
            np.testing.assert_array_equal(ns["v2f"][no[e2]:no[e2 + 1]], state["v2f"][g.msg_off[e]:g.msg_off[e + 1]])
            assert ns["count_v2f"][e2] == state["count_v2f"][e] and ns["count_f2v"][e2] == state["count_f2v"][e]
        assert ns["cycles"] == 9


def check_dynamic_run(oracle_mod, lib_path=None, dtype="f64", seed=1, float_tables=True):
    """A run with every kind of change in it, engine == oracle bit for bit all along: table swap
    (other order of the same variables), scope changes (shrink, grow, replace), external values
    sliced on the device, checkpoint / resume."""
    g = G.random_mixed(40, 60, seed=seed, max_arity=3, dom_choices=(2, 3, 4), names=False,
                       float_tables=float_tables)   # integer tables: classes store narrow records
    p = Params(dtype=dtype, start_messages="leafs_vars")
    rng = np.random.default_rng(seed)

    def factory(graph, params):
        from pydcop_amd.engine import MaxSumEngine
        return MaxSumEngine(graph, params, lib_path=lib_path)
    run = DynamicMaxSum(g, p, engine_factory=factory)
    ora = oracle_mod.OracleMaxSum(g, p)

    def same(step):
        for x, y, what in zip(run.messages(), ora.messages(), ("v2f", "f2v", "cv", "cf")):
            np.testing.assert_array_equal(x, y, err_msg=f"{what} after {step}")
        np.testing.assert_array_equal(run.assignment()[0], ora.assignment()[0], err_msg=step)
        np.testing.assert_array_equal(run.assignment()[1], ora.assignment()[1], err_msg=step)
        a, b = run.eval_cost(), ora.eval_cost()
        assert a[1] == b[1] and abs(a[0] - b[0]) <= 1e-9 * max(1.0, abs(b[0])), step
        assert run.cycle_count == ora.cycle_count

    def oracle_follow():
        """The oracle takes the same new graph + state the product computed: what is checked is
        that the ENGINE carries on from it exactly like the restated algorithm does."""
        nonlocal ora
        st = ora.state()
        return st

    run.run(5), ora.run(5)
    same("5 cycles")
    arities = np.diff(g.factor_rowptr)
    f2 = int(np.flatnonzero(arities == 2)[0])
    f3 = int(np.flatnonzero(arities == 3)[0])
    # 1. same variables, other order
    sc = [int(x) for x in g.edge_var[g.factor_rowptr[f2]:g.factor_rowptr[f2 + 1]]]
    t = rng.integers(-5, 9, [int(g.dom_size[v]) for v in sc[::-1]]).astype(float)
    run.change_factor_function(f2, t, scope=sc[::-1])
    ora.update_factor_table(f2, np.ascontiguousarray(t.T))
    run.run(3), ora.run(3)
    same("table swap, transposed scope")
    # 2. scope changes: shrink an arity-3 factor to one of its variables + a new one, then grow it
    cur_graph = run.graph
    for step, pick in (("shrink+replace", lambda old, free: [free[0], old[1]]),
                       ("grow", lambda old, free: old + [free[1]]),
                       ("unary", lambda old, free: [old[0]])):
        gg = run.graph
        old = [int(x) for x in gg.edge_var[gg.factor_rowptr[f3]:gg.factor_rowptr[f3 + 1]]]
        free = [v for v in range(gg.n_vars) if v not in old]
        scope = pick(old, free)
        tab = rng.uniform(-4, 4, int(np.prod(gg.dom_size[scope])))
        ng, ns = rescope_factor(gg, ora.state(), f3, scope, tab, p.mode, p.dtype)
        run.change_factor_function(f3, tab, scope=scope)
        ora.close()
        ora = oracle_mod.OracleMaxSum(ng, p)
        ora.set_state(**ns)
        same(f"{step}: state right after the change")
        run.run(4), ora.run(4)
        same(step)
    assert run.relayouts == 3
    # 3. external variable: a relation over (x, y, sensor), sliced on the device when the sensor moves
    gg = run.graph
    sc = [int(x) for x in gg.edge_var[gg.factor_rowptr[f2]:gg.factor_rowptr[f2 + 1]]]
    dims = [int(gg.dom_size[sc[0]]), 5, int(gg.dom_size[sc[1]])]   # the sensor sits in the middle
    parent = rng.uniform(-3, 3, dims) if float_tables else rng.integers(-9, 9, dims).astype(float)
    run.register_external(f2, parent, [0, 1, 0])
    for sensor in (3, 0, 4):
        run.set_external_values(f2, [sensor])
        ora.update_factor_table(f2, np.ascontiguousarray(parent[:, sensor, :]))
        run.run(3), ora.run(3)
        same(f"sensor = {sensor}")
    # 4. checkpoint / resume: another engine picks the run up from the saved state
    st = run.engine.state()
    other = factory(run.graph, p)
    other.run(2)                       # (some other state first)
    other.set_state(**st)
    other.run(6), ora.run(6), run.run(6)
    same("after resume (original)")
    np.testing.assert_array_equal(other.assignment()[1], ora.assignment()[1])
    for x, y in zip(other.messages(), ora.messages()):
        np.testing.assert_array_equal(x, y)
    assert other.cycle_count == ora.cycle_count
    # errors
    import pytest
    from pydcop_amd.engine import MaxSumGpuError
    with pytest.raises(MaxSumGpuError):
        run.engine.slice_factor(f2, [7])
    with pytest.raises(MaxSumGpuError):
        run.engine.slice_factor(f3, [0])            # no parent registered
    with pytest.raises(MaxSumGpuError):
        run.engine.set_parent_table(f2, np.zeros((9, 5, 9)), [0, 1, 0])
    with pytest.raises(ValueError):
        run.change_factor_function(f2, np.zeros(4), scope=[0, 0])
    other.close(), run.close(), ora.close()
