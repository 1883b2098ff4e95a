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

    Cost: the carry-over walks the edges and variables in Python and `DynamicMaxSum` then builds a
    fresh engine (layout + upload) -- seconds of host time at 100k..1M variables, paid under the
    plug-in's session lock.  A change of scope is a rare event in the reference's model (a rule
    rewritten, maxsum_dynamic.py:234-271); same-scope changes and external-value moves -- the frequent
    ones -- are device-side kernels and cost microseconds."""
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
    var_lists = []
    for v in range(g.n_vars):
        k0, k1 = int(g.var_rowptr[v]), int(g.var_rowptr[v + 1])
        lst = [int(old_to_new[e]) for e in g.var_edges[k0:k1] if old_to_new[e] >= 0]
        var_lists.append(lst)
    for i, v in enumerate(new_scope):
        if v in added:
            var_lists[v].append(e0 + i)                        # `self._factors.append(factor_name)`
    var_rowptr = np.zeros(g.n_vars + 1, dtype=np.int32)
    np.cumsum([len(x) for x in var_lists], out=var_rowptr[1:])
    var_edges = np.array([e for lst in var_lists for e in lst], dtype=np.int32)

    ng = FlatGraph(dom_size=g.dom_size, var_cost=g.var_cost, factor_rowptr=factor_rowptr.astype(np.int32),
                   edge_var=edge_var, table_off=table_off, tables=tables, var_rowptr=var_rowptr,
                   var_edges=var_edges, init_idx=g.init_idx, eval_var_cost=g.eval_var_cost,
                   var_names=g.var_names, factor_names=g.factor_names, domains=g.domains).validate()

    # ---- state ---------------------------------------------------------------------------
    old_off, new_off = g.msg_off, ng.msg_off
    nm = int(new_off[-1])
    v2f, f2v = np.zeros(nm), np.zeros(nm)
    cv, cf = np.zeros(ng.n_edges, dtype=np.uint8), np.zeros(ng.n_edges, dtype=np.uint8)
    for e_new in range(ng.n_edges):  # O(E) copies (a vectorised gather would do for big graphs)
        e_old = int(new_edge_old[e_new])
        if e_old < 0:
            continue
        D = int(new_off[e_new + 1] - new_off[e_new])
        v2f[new_off[e_new]:new_off[e_new] + D] = state["v2f"][old_off[e_old]:old_off[e_old] + D]
        f2v[new_off[e_new]:new_off[e_new] + D] = state["f2v"][old_off[e_old]:old_off[e_old] + D]
        cv[e_new], cf[e_new] = state["count_v2f"][e_old], state["count_f2v"][e_old]
    # REMOVE at the variable (maxsum_dynamic.py:360-390): `self._prev_messages.clear()`
    for v in removed:
        for e_new in var_lists[v]:
            cv[e_new] = 0
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
