"""Instance compiler: pyDCOP objects -> FlatGraph (indices only).

Works on the reference's own objects by duck typing (nothing is imported from
pydcop): `Variable` (pydcop/dcop/objects.py:175), `Constraint`
(pydcop/dcop/relations.py:672 NAryMatrixRelation, :456 NAryFunctionRelation),
`FactorComputationNode` / `VariableComputationNode`
(pydcop/computations_graph/factor_graph.py:45,104).
"""
import itertools
from typing import Iterable, List, Optional

import numpy as np

from .graph import FlatGraph


_TABLE_CACHE = {}       # (code of the cost function, domains in scope order) -> table
_TABLE_CACHE_MAX = 4096
_VERIFY_ALL_BELOW = 4096   # vectorised tables up to this size are checked entry by entry


def _function_key(constraint, dims):
    """Identity of an intentional constraint up to the NAMES of its variables: two
    constraints whose cost functions have the same code, the same constants and
    globals, the same fixed parameters and the same domains in scope order have the
    same table (a 200 000-factor colouring instance has one `1 if a == b else 0`)."""
    f = getattr(constraint, "_f", None)
    fn = getattr(f, "exp_func", None)
    code = getattr(fn, "__code__", None)
    if code is None or getattr(fn, "__closure__", None):
        return None
    try:
        argnames = code.co_varnames[:code.co_argcount]
        pos = {v.name: i for i, v in enumerate(dims)}
        if set(argnames) != set(pos) or getattr(f, "_fixed_vars", None):
            return None
        # scope position of every argument: the function sees the domains in that order
        arg_pos = tuple(pos[a] for a in argnames)
        doms = tuple(tuple(v.domain) for v in dims)
        return (code.co_code, code.co_consts, code.co_names, arg_pos, doms,
                getattr(f, "_source_file", None))
    except TypeError:  # unhashable domain values
        return None


def _vectorised_table(constraint, dims, shape):
    """Evaluate an expression-based constraint on broadcast numpy arrays of the domain
    values (one call instead of one per entry).  Returns None when the expression
    does not vectorise (conditional expressions, non-numeric domains, ...); the result
    is checked against scalar evaluation on sampled entries before it is trusted."""
    f = getattr(constraint, "_f", None)
    fn = getattr(f, "exp_func", None)
    if fn is None or getattr(f, "_fixed_vars", None):
        return None
    try:
        grids = {}
        for i, v in enumerate(dims):
            vals = np.asarray(list(v.domain))
            if vals.dtype.kind not in "iuf":
                return None
            sh = [1] * len(dims)
            sh[i] = len(vals)
            grids[v.name] = vals.reshape(sh).astype(np.float64 if vals.dtype.kind == "f" else np.int64)
        with np.errstate(all="ignore"):
            out = fn(**grids)
        out = np.broadcast_to(np.asarray(out, dtype=np.float64), shape).copy()
    except Exception:
        return None
    # trust, but verify through the scalar path: EVERY entry of a small table; for a large one
    # the corners, pseudo-random entries and every entry where numpy and Python arithmetic are
    # known to part (non-finite results: Python raises ZeroDivisionError / OverflowError where
    # numpy yields inf or nan; magnitudes beyond 2^53: int64 wraps where Python ints do not)
    names = [v.name for v in dims]
    size = int(np.prod(shape))
    if size <= _VERIFY_ALL_BELOW:
        probe = range(size)
    else:
        flat = out.reshape(-1)
        suspect = np.flatnonzero(~np.isfinite(flat) | (np.abs(flat) >= 2.0 ** 53))
        if suspect.shape[0] > 256:
            return None
        probe = {0, size - 1} | {(k * 2654435761) % size for k in range(1, 64)} | {int(i) for i in suspect}
    for lin in probe:
        idx = np.unravel_index(lin, shape)
        try:
            want = constraint(**{n: dims[i].domain[int(j)] for i, (n, j) in enumerate(zip(names, idx))})
        except ArithmeticError:
            return None  # the scalar path raises here: let the caller's enumeration surface it
        got = out[idx]
        if not (got == want or (np.isnan(got) and want != want)):
            return None
    return out


def tensorise_constraint(constraint) -> np.ndarray:
    """Dense cost tensor of a constraint, row-major over `constraint.dimensions`
    (pydcop/dcop/relations.py:682-690).  Extensional relations expose their
    ndarray.  Intentional ones (NAryFunctionRelation, relations.py:456) are looked up
    in a cache keyed by the function's code and the scope's domains, else evaluated on
    broadcast arrays when the expression allows it, else enumerated once through
    `constraint(**assignment)` -- the reference evaluates them that way on EVERY
    message (relations.py:735-810).  NB: the dimension order of an ExpressionFunction
    constraint is set order, pydcop/utils/expressionfunction.py:74: always read
    `.dimensions`."""
    dims = list(constraint.dimensions)
    shape = tuple(len(v.domain) for v in dims)
    m = getattr(constraint, "_m", None)
    if isinstance(m, np.ndarray) and m.shape == shape:
        return np.ascontiguousarray(m, dtype=np.float64)
    key = _function_key(constraint, dims)
    try:
        if key is not None and key in _TABLE_CACHE:
            return _TABLE_CACHE[key]
    except TypeError:  # an unhashable domain value inside the key: no caching for this one
        key = None
    out = _vectorised_table(constraint, dims, shape)
    if out is None:
        names = [v.name for v in dims]
        out = np.empty(shape, dtype=np.float64)
        flat = out.reshape(-1)
        for i, values in enumerate(itertools.product(*[list(v.domain) for v in dims])):
            flat[i] = constraint(**dict(zip(names, values)))
    if key is not None and len(_TABLE_CACHE) < _TABLE_CACHE_MAX:
        out.setflags(write=False)
        _TABLE_CACHE[key] = out
    return out


def _domain_index(variable, value) -> int:
    for i, d in enumerate(variable.domain):
        if d == value:
            return i
    raise ValueError(f"{value!r} is not in the domain of {variable.name}")


def compile_nodes(var_nodes: Iterable, factor_nodes: Iterable,
                  noise: float = 0.0, rng: Optional[np.random.Generator] = None) -> FlatGraph:
    """Compile factor-graph computation nodes (what `ComputationDef.node`
    carries, pydcop/algorithms/__init__.py:336-380) into a FlatGraph.

    Variable order = order of `var_nodes`; factor order = order of
    `factor_nodes`; a variable's edge order = its node's `links` order, which is
    the order MaxSumVariableComputation iterates its factors
    (pydcop/algorithms/maxsum.py:466, 536).

    `noise` > 0 reproduces VariableNoisyCostFunc (pydcop/dcop/objects.py:547-567):
    one `uniform(0, noise)` draw per domain value added to the variable cost --
    but from a seedable generator instead of the reference's unseeded `random`.
    """
    var_nodes, factor_nodes = list(var_nodes), list(factor_nodes)
    variables = [n.variable for n in var_nodes]
    var_id = {n.name: i for i, n in enumerate(var_nodes)}
    dom_size = np.array([len(v.domain) for v in variables], dtype=np.int32)
    costs = []
    init_idx = np.full(len(variables), -1, dtype=np.int32)
    for i, v in enumerate(variables):
        costs.extend(float(v.cost_for_val(d)) for d in v.domain)
        if getattr(v, "initial_value", None) is not None:
            init_idx[i] = _domain_index(v, v.initial_value)
    var_cost = np.array(costs, dtype=np.float64)
    clean_cost = None
    if noise:
        rng = rng or np.random.default_rng()
        clean_cost = var_cost  # what DCOP.solution_cost sums (dcop.py:352-365): no noise
        var_cost = var_cost + rng.uniform(0.0, noise, size=var_cost.shape)

    factor_rowptr = [0]
    edge_var: List[int] = []
    table_off = [0]
    tables = []
    edge_of = {}  # (factor name, var name) -> edge id
    for fn in factor_nodes:
        factor = fn.factor
        for v in factor.dimensions:
            if v.name not in var_id:
                raise ValueError(f"factor {fn.name} depends on unknown variable {v.name}")
            edge_of[(fn.name, v.name)] = len(edge_var)
            edge_var.append(var_id[v.name])
        factor_rowptr.append(len(edge_var))
        t = tensorise_constraint(factor).reshape(-1)
        tables.append(t)
        table_off.append(table_off[-1] + t.shape[0])

    var_rowptr = [0]
    var_edges: List[int] = []
    for n in var_nodes:
        for link in n.links:
            key = (link.factor_node, n.name)
            if key not in edge_of:
                raise ValueError(f"variable {n.name} is linked to unknown factor {link.factor_node}")
            var_edges.append(edge_of[key])
        var_rowptr.append(len(var_edges))

    g = FlatGraph(
        dom_size=dom_size, var_cost=var_cost,
        factor_rowptr=np.array(factor_rowptr, dtype=np.int32),
        edge_var=np.array(edge_var, dtype=np.int32),
        table_off=np.array(table_off, dtype=np.int64),
        tables=np.concatenate(tables) if tables else np.zeros(0),
        var_rowptr=np.array(var_rowptr, dtype=np.int32),
        var_edges=np.array(var_edges, dtype=np.int32),
        init_idx=init_idx if (init_idx >= 0).any() else None,
        eval_var_cost=clean_cost,
        var_names=[n.name for n in var_nodes],
        factor_names=[n.name for n in factor_nodes],
        domains=[list(v.domain) for v in variables],
    )
    return g.validate()


def compile_computation_graph(cg, noise: float = 0.0, rng=None) -> FlatGraph:
    """Compile a ComputationsFactorGraph
    (pydcop/computations_graph/factor_graph.py:210)."""
    var_nodes = [n for n in cg.nodes if n.type == "VariableComputation"]
    factor_nodes = [n for n in cg.nodes if n.type == "FactorComputation"]
    return compile_nodes(var_nodes, factor_nodes, noise=noise, rng=rng)


def assignment_to_values(graph: FlatGraph, idx: np.ndarray) -> dict:
    """{variable name: domain value} for an index assignment."""
    return {n: graph.domains[i][int(idx[i])] for i, n in enumerate(graph.var_names)}
