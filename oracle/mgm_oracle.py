"""ctypes wrapper of oracle/mgm_oracle.c -- TEST INFRASTRUCTURE ONLY (never imported by pydcop_amd/)."""
import ctypes as C
import os

import numpy as np

from pydcop_amd.graph import CGraph, CParams, FlatGraph, Params

from .maxsum_oracle import build

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def _lib(dtype):
    if dtype not in _LIBS:
        path = os.path.join(_HERE, f"libmgm_oracle_{dtype}.so")
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        vp = C.c_void_p
        lib.mgmo_create.restype = vp
        lib.mgmo_create.argtypes = [C.POINTER(CGraph), C.POINTER(CParams), vp]
        lib.mgmo_reset.argtypes = [vp]
        lib.mgmo_set_value_rank.argtypes = [vp, vp]
        lib.mgmo_run.argtypes = [vp, C.c_int32]
        lib.mgmo_rounds.restype = C.c_int64
        lib.mgmo_rounds.argtypes = [vp]
        lib.mgmo_get_state.argtypes = [vp] + [vp] * 5
        lib.mgmo_eval_cost.argtypes = [vp, vp, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        lib.mgmo_destroy.argtypes = [vp]
        _LIBS[dtype] = lib
    return _LIBS[dtype]


def name_ranks(names):
    """Rank of every variable's name in sorted order (the reference breaks ties with sorted(names))."""
    order = sorted(range(len(names)), key=lambda i: names[i])
    rank = np.empty(len(names), dtype=np.int32)
    rank[order] = np.arange(len(names), dtype=np.int32)
    return rank


class OracleMgm:
    """Same surface as pydcop_amd.mgm.MgmEngine."""

    def __init__(self, graph: FlatGraph, params: Params = None):
        self.graph = graph
        self.params = params or Params()
        self._lib = _lib(self.params.dtype)
        cg, cp = graph.to_c(), self.params.to_c()
        self._rank = name_ranks(graph.var_names) if graph.var_names else None
        self._h = self._lib.mgmo_create(C.byref(cg), C.byref(cp),
                                        None if self._rank is None else self._rank.ctypes.data)
        self._vrank = graph.value_rank()   # the order of the domain values (cost ties at the start)
        if self._vrank is not None:
            self._lib.mgmo_set_value_rank(self._h, self._vrank.ctypes.data)

    def reset(self):
        self._lib.mgmo_reset(self._h)

    def run(self, n_rounds: int):
        self._lib.mgmo_run(self._h, int(n_rounds))

    @property
    def cycle_count(self) -> int:
        return int(self._lib.mgmo_rounds(self._h))

    def state(self) -> dict:
        n = self.graph.n_vars
        out = {"idx": np.empty(n, dtype=np.int32), "cost": np.empty(n), "has_cost": np.empty(n, dtype=np.uint8),
               "gain": np.empty(n), "new": np.empty(n, dtype=np.int32)}
        self._lib.mgmo_get_state(self._h, *[out[k].ctypes.data for k in ("idx", "cost", "has_cost", "gain", "new")])
        return out

    def assignment(self):
        s = self.state()
        return s["idx"], s["cost"]

    def eval_cost(self, idx=None, infinity=float("inf")):
        cost, viol = C.c_double(), C.c_int64()
        p = None
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            p = idx.ctypes.data
        self._lib.mgmo_eval_cost(self._h, p, float(infinity), C.byref(cost), C.byref(viol))
        return cost.value, int(viol.value)

    def close(self):
        if self._h:
            self._lib.mgmo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
