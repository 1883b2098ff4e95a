"""Flat ("compiled") factor graph: the arrays the C-ABI takes.

Mirrors `struct mxs_graph` / `struct mxs_params` of include/maxsum_gpu.h.  The
device only ever sees indices; domain *values* stay on the host (the reference
keys its messages by value, pydcop/algorithms/maxsum.py:223-235).
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

MODE = {"min": 0, "max": 1}
DAMPING_NODES = {"none": 0, "vars": 1, "factors": 2, "both": 3}
START_MESSAGES = {"leafs": 0, "leafs_vars": 1, "all": 2}
DTYPE = {"f64": 0, "f32": 1}


class CGraph(C.Structure):
    """`struct mxs_graph` (include/maxsum_gpu.h)."""
    _fields_ = [
        ("n_vars", C.c_int32), ("n_factors", C.c_int32), ("n_edges", C.c_int32),
        ("dom_size", C.c_void_p), ("var_cost", C.c_void_p), ("init_idx", C.c_void_p),
        ("factor_rowptr", C.c_void_p), ("edge_var", C.c_void_p),
        ("table_off", C.c_void_p), ("tables", C.c_void_p),
        ("var_rowptr", C.c_void_p), ("var_edges", C.c_void_p),
        ("var_owned", C.c_void_p), ("factor_owned", C.c_void_p),
        ("eval_var_cost", C.c_void_p),
    ]


class CParams(C.Structure):
    """`struct mxs_params` (include/maxsum_gpu.h)."""
    _fields_ = [
        ("mode", C.c_int32), ("damping_nodes", C.c_int32),
        ("start_messages", C.c_int32), ("dtype", C.c_int32),
        ("damping", C.c_double), ("stability", C.c_double),
        ("graph_chunk", C.c_int32), ("layout_flags", C.c_int32),
    ]


@dataclass
class Params:
    """Algorithm parameters: the reference's `algo_params`
    (pydcop/algorithms/maxsum.py:212-220) minus `noise` (folded into
    `FlatGraph.var_cost` by the compiler) plus engine knobs."""
    mode: str = "min"
    damping: float = 0.5
    damping_nodes: str = "both"
    stability: float = 0.1
    start_messages: str = "leafs"
    dtype: str = "f64"
    graph_chunk: int = -1
    layout_flags: int = 0

    def to_c(self) -> CParams:
        for name, table in (("mode", MODE), ("damping_nodes", DAMPING_NODES),
                            ("start_messages", START_MESSAGES), ("dtype", DTYPE)):
            if getattr(self, name) not in table:
                raise ValueError(
                    f"Invalid value {getattr(self, name)!r} for parameter {name}, "
                    f"must be one of {sorted(table)}")
        return CParams(MODE[self.mode], DAMPING_NODES[self.damping_nodes],
                       START_MESSAGES[self.start_messages], DTYPE[self.dtype],
                       float(self.damping), float(self.stability),
                       int(self.graph_chunk), int(self.layout_flags))


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


@dataclass
class FlatGraph:
    """Factor graph in the C-ABI array format (see include/maxsum_gpu.h)."""
    dom_size: np.ndarray        # int32 [n_vars]
    var_cost: np.ndarray        # float64 [sum dom_size]
    factor_rowptr: np.ndarray   # int32 [n_factors+1]
    edge_var: np.ndarray        # int32 [n_edges]
    table_off: np.ndarray       # int64 [n_factors+1]
    tables: np.ndarray          # float64 [table_off[-1]]
    var_rowptr: np.ndarray      # int32 [n_vars+1]
    var_edges: np.ndarray       # int32 [n_edges]
    init_idx: Optional[np.ndarray] = None   # int32 [n_vars] or None
    var_owned: Optional[np.ndarray] = None  # uint8 [n_vars] or None
    factor_owned: Optional[np.ndarray] = None  # uint8 [n_factors] or None
    # float64 [sum dom_size] or None: the variables' own costs, WITHOUT the Max-Sum noise that
    # compile_nodes folds into var_cost -- what the solution cost is evaluated on
    # (DCOP.solution_cost, pydcop/dcop/dcop.py:308-367, never sees the noise)
    eval_var_cost: Optional[np.ndarray] = None
    # host-only metadata (names / domain values), optional
    var_names: Optional[List[str]] = None
    factor_names: Optional[List[str]] = None
    domains: Optional[List[Sequence]] = None
    _keep: list = field(default_factory=list, repr=False)

    def __post_init__(self):
        self.dom_size = _arr(self.dom_size, np.int32)
        self.var_cost = _arr(self.var_cost, np.float64)
        self.factor_rowptr = _arr(self.factor_rowptr, np.int32)
        self.edge_var = _arr(self.edge_var, np.int32)
        self.table_off = _arr(self.table_off, np.int64)
        self.tables = _arr(self.tables, np.float64)
        self.var_rowptr = _arr(self.var_rowptr, np.int32)
        self.var_edges = _arr(self.var_edges, np.int32)
        if self.init_idx is not None:
            self.init_idx = _arr(self.init_idx, np.int32)
        if self.var_owned is not None:
            self.var_owned = _arr(self.var_owned, np.uint8)
        if self.factor_owned is not None:
            self.factor_owned = _arr(self.factor_owned, np.uint8)
        if self.eval_var_cost is not None:
            self.eval_var_cost = _arr(self.eval_var_cost, np.float64)

    # sizes ---------------------------------------------------------------
    @property
    def n_vars(self) -> int:
        return int(self.dom_size.shape[0])

    @property
    def n_factors(self) -> int:
        return int(self.factor_rowptr.shape[0] - 1)

    @property
    def n_edges(self) -> int:
        return int(self.edge_var.shape[0])

    @property
    def cost_off(self) -> np.ndarray:
        off = np.zeros(self.n_vars + 1, dtype=np.int64)
        np.cumsum(self.dom_size, out=off[1:])
        return off

    def value_rank(self) -> Optional[np.ndarray]:
        """rank[cost_off[v] + d] = position of the d-th value of v's domain among v's values in ascending
        (Python) order, or None when every domain is written in ascending order / no domain values are
        known.  What the reference's `optimal_cost_value` breaks cost ties on (relations.py:1661-1665:
        min / max over (cost, value) tuples) -- DSA and MGM start a variable without neighbours there.
        Values that do not compare (or cannot be hashed): the reference's tuple comparison would raise on a
        cost tie; here a warning is logged once and the INDEX order breaks such ties."""
        if self.domains is None:
            return None
        rank = np.empty(int(self.cost_off[-1]), dtype=np.int32)
        off, identity, seen = 0, True, {}
        for values in self.domains:
            try:
                key = tuple(values)        # (instances share a few domains among many variables)
                r = seen.get(key)
                if r is None:
                    order = sorted(range(len(key)), key=lambda d: key[d])
            except TypeError:
                import logging
                logging.getLogger("pydcop_amd").warning(
                    "domain values %r neither compare nor hash: cost ties of variables without neighbours "
                    "(DSA / MGM start values) are broken on the value's INDEX, where the reference compares "
                    "the values themselves (relations.py:1661-1665)", list(values)[:4])
                return None
            if r is None:
                r = np.empty(len(key), dtype=np.int32)
                r[order] = np.arange(len(key), dtype=np.int32)
                seen[key] = r
                identity = identity and bool((r == np.arange(len(key))).all())
            rank[off:off + len(key)] = r
            off += len(key)
        return None if identity else rank

    @property
    def msg_off(self) -> np.ndarray:
        """Offset of edge e's message in the message arrays returned by
        `get_messages` (exclusive prefix sum of the edge domain sizes)."""
        off = np.zeros(self.n_edges + 1, dtype=np.int64)
        np.cumsum(self.dom_size[self.edge_var], out=off[1:])
        return off

    # construction helpers -------------------------------------------------
    @staticmethod
    def var_side_from_edges(edge_var: np.ndarray, n_vars: int):
        """Variable-side CSR with each variable's edges in increasing edge id,
        i.e. in the order of the factors -- which is the links order the
        reference's build_computation_graph produces
        (pydcop/computations_graph/factor_graph.py:276-280)."""
        edge_var = np.asarray(edge_var, dtype=np.int64)
        order = np.argsort(edge_var, kind="stable").astype(np.int32)
        counts = np.bincount(edge_var, minlength=n_vars)
        rowptr = np.zeros(n_vars + 1, dtype=np.int32)
        np.cumsum(counts, out=rowptr[1:])
        return rowptr, order

    def validate(self):
        nv, nf, ne = self.n_vars, self.n_factors, self.n_edges
        if (self.dom_size < 1).any():
            raise ValueError("every variable needs a non-empty domain")
        if self.var_cost.shape[0] != int(self.dom_size.sum()):
            raise ValueError("var_cost must hold sum(dom_size) entries")
        if self.eval_var_cost is not None and self.eval_var_cost.shape != self.var_cost.shape:
            raise ValueError("eval_var_cost must have the shape of var_cost")
        if self.factor_rowptr[0] != 0 or self.factor_rowptr[-1] != ne:
            raise ValueError("factor_rowptr does not span the edges")
        if (np.diff(self.factor_rowptr) < 1).any():
            raise ValueError("every factor needs at least one variable")
        if ne and (self.edge_var.min() < 0 or self.edge_var.max() >= nv):
            raise ValueError("edge_var out of range")
        if self.var_rowptr[0] != 0 or self.var_rowptr[-1] != ne:
            raise ValueError("var_rowptr does not span the edges")
        if ne:
            if not np.array_equal(np.sort(self.var_edges), np.arange(ne, dtype=np.int32)):
                raise ValueError("var_edges must be a permutation of the edges")
            owner = np.repeat(np.arange(nv, dtype=np.int32), np.diff(self.var_rowptr))
            if not np.array_equal(self.edge_var[self.var_edges], owner):
                raise ValueError("var_edges inconsistent with edge_var")
        sizes = np.ones(nf, dtype=np.int64)
        d = self.dom_size[self.edge_var].astype(np.int64)
        if nf:
            sizes = np.multiply.reduceat(d, self.factor_rowptr[:-1].astype(np.int64))
        if not np.array_equal(np.diff(self.table_off), sizes):
            raise ValueError("table_off inconsistent with the factor scopes")
        if self.tables.shape[0] != int(self.table_off[-1]):
            raise ValueError("tables has the wrong size")
        return self

    def to_c(self) -> CGraph:
        def p(a):
            return None if a is None else a.ctypes.data_as(C.c_void_p)
        return CGraph(self.n_vars, self.n_factors, self.n_edges,
                      p(self.dom_size), p(self.var_cost), p(self.init_idx),
                      p(self.factor_rowptr), p(self.edge_var), p(self.table_off),
                      p(self.tables), p(self.var_rowptr), p(self.var_edges),
                      p(self.var_owned), p(self.factor_owned), p(self.eval_var_cost))

    # compact binary instance format (.npz) beside YAML --------------------------
    _ARRAYS = ("dom_size", "var_cost", "factor_rowptr", "edge_var", "table_off", "tables",
               "var_rowptr", "var_edges")
    _OPTIONAL = ("init_idx", "var_owned", "factor_owned", "eval_var_cost")

    def save(self, path: str, objective: str = "min", **meta):
        """Write the compiled instance as one compressed .npz: the flat arrays of the
        C-ABI plus a JSON header (objective, names, domain values, free-form `meta`).
        A 100k-variable instance that takes pyDCOP minutes to parse from YAML
        (pydcop/dcop/yamldcop.py:96) and compile loads back in milliseconds."""
        import json
        header = {"format": "maxsum_gpu.flatgraph", "version": 1, "objective": objective,
                  "var_names": self.var_names, "factor_names": self.factor_names,
                  "domains": [list(d) for d in self.domains] if self.domains is not None else None,
                  "meta": meta}
        arrays = {k: getattr(self, k) for k in self._ARRAYS}
        for k in self._OPTIONAL:
            if getattr(self, k) is not None:
                arrays[k] = getattr(self, k)
        arrays["header"] = np.frombuffer(
            json.dumps(header, default=lambda o: o.item() if hasattr(o, "item") else str(o)).encode(),
            dtype=np.uint8)
        np.savez_compressed(path, **arrays)

    @classmethod
    def load(cls, path: str):
        """-> (FlatGraph, header dict) from a file written by `save`; validated."""
        import json
        z = np.load(path)
        if "header" not in z.files:
            raise ValueError(f"{path}: not a maxsum_gpu instance file (no header)")
        header = json.loads(bytes(z["header"]).decode())
        if header.get("format") != "maxsum_gpu.flatgraph" or header.get("version") != 1:
            raise ValueError(f"{path}: unsupported instance format {header.get('format')!r} "
                             f"version {header.get('version')!r}")
        g = cls(**{k: z[k] for k in cls._ARRAYS},
                **{k: z[k] for k in cls._OPTIONAL if k in z.files})
        g.var_names, g.factor_names = header.get("var_names"), header.get("factor_names")
        g.domains = header.get("domains")
        return g.validate(), header

    # algorithmic bytes of one cycle -----------------
    def cycle_bytes(self, word: int) -> int:
        d_e = self.dom_size[self.edge_var].astype(np.int64)
        b = int((6 * d_e * word + 8).sum())
        b += int(np.diff(self.table_off).sum()) * word
        b += int((self.dom_size.astype(np.int64) * word + 8 + word).sum())
        return b
