"""Dynamic Max-Sum on the GPU engine: a run that survives changes of the factor graph itself.

Restates `pydcop/algorithms/maxsum_dynamic.py` (whose computation classes are marked broken since
the refactoring of maxsum -- maxsum_dynamic.py:60 -- and have no counterpart in the reference's
tests beyond the same-scope `change_factor_function`, tests/unit/test_algorithms_dynamic_maxsum.py):

  * `change_factor_function`, same scope (:80-104)            -> `mxs_update_factor_table`
  * a relation that also depends on external / read-only variables, re-sliced when one of them
    changes value (:113-186, :188-232, :273-288)              -> `mxs_set_parent_table` +
                                                                 `mxs_slice_factor` (device-side)
  * `change_factor_function` with a NEW scope (:234-271): the factor forgets what the removed
    variables sent and what it sent them, takes all-zero costs for the added ones and sends
    them its costs (ADD, :290-313); a removed variable drops the factor, what it held from
    it, and ALL its previous messages (REMOVE, :360-390); an added variable appends the factor
    to its list and holds the ADD costs (:392-398)           -> `rescope_factor` below: the
    new flat graph + the carried-over state, loaded into a fresh engine with `mxs_set_state`.

Everything else of a cycle is the synchronous Max-Sum of the engine.
"""
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from .graph import FlatGraph, Params


def _factor_message(table: np.ndarray, msgs: Sequence[np.ndarray], pos: int, mode: str, dtype) -> np.ndarray:
    """factor_costs_for_var (pydcop/algorithms/maxsum.py:382-447) for scope position `pos`:
    opt over the other variables of  f_val + sum_cost,  sum_cost = ((0 + m_a) + m_b) + ... in
    dimensions order -- the reference's order of additions, element for element."""
    ar = table.ndim
    sum_cost = np.zeros((1,) * ar, dtype=dtype)
    for o in range(ar):
        if o == pos:
            continue
        shape = [1] * ar
        shape[o] = table.shape[o]
        sum_cost = sum_cost + np.asarray(msgs[o], dtype=dtype).reshape(shape)
    total = table.astype(dtype) + sum_cost
    axes = tuple(a for a in range(ar) if a != pos)
    if not axes:
        return np.asarray(total, dtype=dtype).reshape(-1)
    return (total.min(axis=axes) if mode == "min" else total.max(axis=axes)).astype(dtype)


def rescope_factor(graph: FlatGraph, state: Dict, factor: int, new_scope: Sequence[int], new_table,
                   mode: str = "min", dtype: str = "f64") -> Tuple[FlatGraph, Dict]:
    """A new cost function with a new scope for `factor` (maxsum_dynamic.py:234-271).

    -> (new FlatGraph, new state) with the semantics of the module docstring: surviving edges
    keep their messages and send counters; an added edge starts with an all-zero V->F message,
    no previous message on either side, and the F->V message the factor sends with ADD; a
    variable that lost the factor has no previous message on any of its remaining edges.

    Cost: array operations over the whole graph (one stable sort of the variable-side links, copies of
    the message and table arrays: 4.5 s on the 1M-variable, 6M-edge instance in the build container;
    round 2 walked every edge and variable in Python), after which `DynamicMaxSum` builds a fresh engine
    (layout + upload) -- seconds of host time at that size, paid under the plug-in's session lock.  A
    change of scope is a rare event in the reference's model (a rule rewritten,
    maxsum_dynamic.py:234-271); same-scope changes and external-value moves -- the frequent ones -- are
    device-side kernels and cost microseconds."""
    g = graph
    f = int(factor)
    new_scope = [int(v) for v in new_scope]
    if len(set(new_scope)) != len(new_scope) or not new_scope:
        raise ValueError("a scope lists each variable once")
    if any(v < 0 or v >= g.n_vars for v in new_scope):
        raise ValueError("variable out of range")
    shape = tuple(int(g.dom_size[v]) for v in new_scope)
    new_table = np.ascontiguousarray(new_table, dtype=np.float64)
    if new_table.size != int(np.prod(shape)):
        raise ValueError("the table does not have the shape of the new scope")
    new_table = new_table.reshape(shape)
    np_dtype = np.float64 if dtype == "f64" else np.float32

    e0, e1 = int(g.factor_rowptr[f]), int(g.factor_rowptr[f + 1])
    old_scope = [int(x) for x in g.edge_var[e0:e1]]
    old_edge_of = {v: e0 + i for i, v in enumerate(old_scope)}
    removed = [v for v in old_scope if v not in new_scope]
    added = [v for v in new_scope if v not in old_scope]

    # ---- factor side: edges are factor-major; only f's block changes ---------------------
    nE_old = g.n_edges
    shift = len(new_scope) - len(old_scope)
    new_edge_old = np.concatenate([np.arange(0, e0), np.array([old_edge_of.get(v, -1) for v in new_scope], dtype=np.int64),
                                   np.arange(e1, nE_old)]).astype(np.int64)       # new edge -> old edge / -1
    old_to_new = np.full(nE_old, -1, dtype=np.int64)
    ok = new_edge_old >= 0
    old_to_new[new_edge_old[ok]] = np.flatnonzero(ok)
    edge_var = np.concatenate([g.edge_var[:e0], np.array(new_scope, dtype=np.int32), g.edge_var[e1:]]).astype(np.int32)
    factor_rowptr = g.factor_rowptr.astype(np.int64).copy()
    factor_rowptr[f + 1:] += shift
    t0, t1 = int(g.table_off[f]), int(g.table_off[f + 1])
    tables = np.concatenate([g.tables[:t0], new_table.reshape(-1), g.tables[t1:]])
    table_off = g.table_off.copy()
    table_off[f + 1:] += new_table.size - (t1 - t0)

    # ---- variable side: links order kept, removed edges dropped, added edges appended ------
    # (array operations over the whole graph: one stable sort; a scope change on a 1M-variable instance
    # used to walk every variable and every edge in Python)
    deg_old = np.diff(g.var_rowptr.astype(np.int64))
    var_of_slot = np.repeat(np.arange(g.n_vars, dtype=np.int64), deg_old)
    slot_new_edge = old_to_new[g.var_edges]
    keep = slot_new_edge >= 0
    add_pos = [i for i, v in enumerate(new_scope) if v in added]
    slot_var = np.concatenate([var_of_slot[keep], np.array([new_scope[i] for i in add_pos], dtype=np.int64)])
    slot_edge = np.concatenate([slot_new_edge[keep], np.array([e0 + i for i in add_pos], dtype=np.int64)])
    order = np.argsort(slot_var, kind="stable")   # kept links in their order, `self._factors.append(factor_name)` last
    var_edges = slot_edge[order].astype(np.int32)
    var_rowptr = np.zeros(g.n_vars + 1, dtype=np.int32)
    np.cumsum(np.bincount(slot_var, minlength=g.n_vars), out=var_rowptr[1:])

    ng = FlatGraph(dom_size=g.dom_size, var_cost=g.var_cost, factor_rowptr=factor_rowptr.astype(np.int32),
                   edge_var=edge_var, table_off=table_off, tables=tables, var_rowptr=var_rowptr,
                   var_edges=var_edges, init_idx=g.init_idx, eval_var_cost=g.eval_var_cost,
                   var_names=g.var_names, factor_names=g.factor_names, domains=g.domains).validate()

    # ---- state ---------------------------------------------------------------------------
    old_off, new_off = g.msg_off, ng.msg_off
    # Only f's block of the factor-major edge order changes: everything before it is carried over as it
    # is, everything after it shifted; inside, a kept variable's messages and counters move to its new
    # position, an added variable starts from zeros.
    m0, m1 = int(old_off[e0]), int(old_off[e1])
    n0, n1 = int(new_off[e0]), int(new_off[e0 + len(new_scope)])

    def carry(old, block):
        return np.concatenate([old[:m0], block, old[m1:]])

    bv, bf = np.zeros(n1 - n0), np.zeros(n1 - n0)
    bcv, bcf = np.zeros(len(new_scope), dtype=np.uint8), np.zeros(len(new_scope), dtype=np.uint8)
    for i, v in enumerate(new_scope):  # (the factor's own edges: its arity)
        e_old = old_edge_of.get(v, -1)
        if e_old < 0:
            continue
        a, b = int(new_off[e0 + i]) - n0, int(new_off[e0 + i + 1]) - n0
        bv[a:b] = state["v2f"][old_off[e_old]:old_off[e_old + 1]]
        bf[a:b] = state["f2v"][old_off[e_old]:old_off[e_old + 1]]
        bcv[i], bcf[i] = state["count_v2f"][e_old], state["count_f2v"][e_old]
    v2f, f2v = carry(np.asarray(state["v2f"], dtype=np.float64), bv), carry(np.asarray(state["f2v"], dtype=np.float64), bf)
    cv = np.concatenate([np.asarray(state["count_v2f"], dtype=np.uint8)[:e0], bcv, np.asarray(state["count_v2f"], dtype=np.uint8)[e1:]])
    cf = np.concatenate([np.asarray(state["count_f2v"], dtype=np.uint8)[:e0], bcf, np.asarray(state["count_f2v"], dtype=np.uint8)[e1:]])
    assert v2f.size == int(new_off[-1]) and cv.size == ng.n_edges
    # REMOVE at the variable (maxsum_dynamic.py:360-390): `self._prev_messages.clear()`
    for v in removed:
        cv[var_edges[var_rowptr[v]:var_rowptr[v + 1]]] = 0
    # ADD (:290-313, 392-398): the factor's costs for the new variable, from what it holds now
    # (`self._costs[v.name] = {d: 0 ...}` for the added ones); nothing was sent before
    msgs = [v2f[new_off[e0 + i]:new_off[e0 + i + 1]] for i in range(len(new_scope))]
    sign_table = new_table
    for i, v in enumerate(new_scope):
        if v in added:
            m = _factor_message(sign_table, msgs, i, mode, np_dtype)
            f2v[new_off[e0 + i]:new_off[e0 + i + 1]] = m.astype(np.float64)
    new_state = {"v2f": v2f, "f2v": f2v, "count_v2f": cv, "count_f2v": cf,
                 "idx": state["idx"], "belief": state["belief"], "cycles": state["cycles"]}
    return ng, new_state


class DynamicMaxSum:
    """A Max-Sum run whose factors can change while it iterates (same surface as `MaxSumEngine`).

    >>> run = DynamicMaxSum(graph, Params())
    >>> run.run(20)
    >>> run.change_factor_function(f, table)                    # same scope: table swapped in place
    >>> run.change_factor_function(f, table, scope=[3, 7, 9])   # new scope: re-layout + state carried over
    >>> run.register_external(f2, parent, is_external=[0, 0, 1]); run.set_external_values(f2, [4])
    """

    def __init__(self, graph: FlatGraph, params: Optional[Params] = None, device: int = 0,
                 lib_path: Optional[str] = None, engine_factory=None):
        self.params = params or Params()
        self._device, self._lib_path = device, lib_path
        self._factory = engine_factory or self._default_factory
        self.graph = graph
        self.engine = self._factory(graph, self.params)
        self._parents = {}
        self.relayouts = 0

    def _default_factory(self, graph, params):
        from .engine import MaxSumEngine
        return MaxSumEngine(graph, params, device=self._device, lib_path=self._lib_path)

    # -- the engine's surface ------------------------------------------------------------------
    def run(self, n_cycles: int):
        self.engine.run(n_cycles)

    def assignment(self):
        return self.engine.assignment()

    def messages(self):
        return self.engine.messages()

    def eval_cost(self, idx=None, infinity: float = float("inf")):
        return self.engine.eval_cost(idx, infinity)

    @property
    def cycle_count(self) -> int:
        return self.engine.cycle_count

    def close(self):
        self.engine.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- changes -------------------------------------------------------------------------------
    def change_factor_function(self, factor: int, table, scope: Optional[Sequence[int]] = None):
        """New cost function for `factor`: `table` row-major over `scope` (variable ids; default:
        the factor's current scope).  Same variables (in any order): the table is swapped on the
        device and the iteration carries on (maxsum_dynamic.py:80-104).  Other variables: the
        scope change of maxsum_dynamic.py:234-271, see `rescope_factor`."""
        g = self.graph
        e0, e1 = int(g.factor_rowptr[factor]), int(g.factor_rowptr[factor + 1])
        old = [int(x) for x in g.edge_var[e0:e1]]
        scope = old if scope is None else [int(v) for v in scope]
        if sorted(scope) == sorted(old):
            if len(set(scope)) != len(scope):
                raise ValueError("a scope lists each variable once")
            t = np.ascontiguousarray(table, dtype=np.float64).reshape([int(g.dom_size[v]) for v in scope])
            t = np.transpose(t, [scope.index(v) for v in old])
            self.engine.update_factor_table(int(factor), np.ascontiguousarray(t))
            return
        ng, ns = rescope_factor(g, self.engine.state(), factor, scope, table, self.params.mode, self.params.dtype)
        new_engine = self._factory(ng, self.params)
        new_engine.set_state(**ns)
        self.engine.close()
        self.engine, self.graph = new_engine, ng
        self._parents.pop(int(factor), None)
        for f, (parent, ext) in self._parents.items():   # parent relations live in the engine
            new_engine.set_parent_table(f, parent, ext)
        self.relayouts += 1

    def register_external(self, factor: int, parent, is_external):
        """`factor`'s whole relation, over its scope AND external (read-only) variables
        (maxsum_dynamic.py:113-186): kept on the device, sliced there by `set_external_values`."""
        parent = np.ascontiguousarray(parent, dtype=np.float64)
        ext = np.ascontiguousarray(is_external, dtype=bool)
        self.engine.set_parent_table(int(factor), parent, ext)
        self._parents[int(factor)] = (parent, ext)

    def set_external_values(self, factor: int, value_idx: Sequence[int]):
        """The external variables of `factor` took these values (indices in their domains, in the
        order of the external dimensions): the factor now optimises the slice
        (`_on_new_var_value_msg`, maxsum_dynamic.py:273-288)."""
        self.engine.slice_factor(int(factor), value_idx)
