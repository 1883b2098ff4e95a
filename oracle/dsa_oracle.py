"""ctypes wrapper of oracle/dsa_oracle.c -- TEST INFRASTRUCTURE ONLY (never imported by pydcop_amd/)."""
import ctypes as C
import os

import numpy as np

from pydcop_amd.graph import CGraph, CParams, FlatGraph, Params

from .maxsum_oracle import build

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}
VARIANTS = {"A": 0, "B": 1, "C": 2}


def _lib(dtype):
    if dtype not in _LIBS:
        path = os.path.join(_HERE, f"libdsa_oracle_{dtype}.so")
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        vp = C.c_void_p
        lib.dsao_create.restype = vp
        lib.dsao_create.argtypes = [C.POINTER(CGraph), C.POINTER(CParams), C.c_int32, C.c_double, C.c_int32, C.c_uint64]
        lib.dsao_reset.argtypes = [vp]
        lib.dsao_set_value_rank.argtypes = [vp, vp]
        lib.dsao_run.argtypes = [vp, C.c_int32]
        lib.dsao_cycles.restype = C.c_int64
        lib.dsao_cycles.argtypes = [vp]
        lib.dsao_get_state.argtypes = [vp, vp, vp]
        lib.dsao_eval_cost.argtypes = [vp, vp, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        lib.dsao_destroy.argtypes = [vp]
        lib.dsa_uniform.restype = C.c_double
        lib.dsa_uniform.argtypes = [C.c_uint64, C.c_int32, C.c_int64, C.c_int32]
        _LIBS[dtype] = lib
    return _LIBS[dtype]


def uniform(seed, variable, cycle, draw):
    return _lib("f64").dsa_uniform(seed, variable, cycle, draw)


class OracleDsa:
    """Same surface as pydcop_amd.dsa.DsaEngine."""

    def __init__(self, graph: FlatGraph, params: Params = None, variant="B", probability=0.7, p_mode="fixed", seed=0):
        self.graph = graph
        self.params = params or Params()
        self._lib = _lib(self.params.dtype)
        cg, cp = graph.to_c(), self.params.to_c()
        self._h = self._lib.dsao_create(C.byref(cg), C.byref(cp), VARIANTS[variant], float(probability),
                                        1 if p_mode == "arity" else 0, int(seed))
        self._vrank = graph.value_rank()   # the order of the domain values (cost ties at the start)
        if self._vrank is not None:
            self._lib.dsao_set_value_rank(self._h, self._vrank.ctypes.data)

    def reset(self):
        self._lib.dsao_reset(self._h)

    def run(self, n_cycles: int):
        self._lib.dsao_run(self._h, int(n_cycles))

    @property
    def cycle_count(self) -> int:
        return int(self._lib.dsao_cycles(self._h))

    def assignment(self):
        idx = np.empty(self.graph.n_vars, dtype=np.int32)
        cost = np.empty(self.graph.n_vars)
        self._lib.dsao_get_state(self._h, idx.ctypes.data, cost.ctypes.data)
        return idx, cost

    def eval_cost(self, idx=None, infinity=float("inf")):
        cost, viol = C.c_double(), C.c_int64()
        p = None
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            p = idx.ctypes.data
        self._lib.dsao_eval_cost(self._h, p, float(infinity), C.byref(cost), C.byref(viol))
        return cost.value, int(viol.value)

    def close(self):
        if self._h:
            self._lib.dsao_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
