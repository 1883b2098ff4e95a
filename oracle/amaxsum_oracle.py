"""ctypes wrapper of oracle/amaxsum_oracle.c -- TEST INFRASTRUCTURE ONLY (never imported by
pydcop_amd/): the reference's asynchronous Max-Sum under FIFO delivery, on the CPU."""
import ctypes as C
import os

import numpy as np

from pydcop_amd.graph import CGraph, CParams, FlatGraph, Params

from .maxsum_oracle import build

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def _lib(dtype):
    if dtype not in _LIBS:
        path = os.path.join(_HERE, f"libamaxsum_oracle_{dtype}.so")
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        vp = C.c_void_p
        lib.amso_create.restype = vp
        lib.amso_create.argtypes = [C.POINTER(CGraph), C.POINTER(CParams)]
        lib.amso_reset.argtypes = [vp]
        lib.amso_run.restype = C.c_int64
        lib.amso_run.argtypes = [vp, C.c_int32, C.c_int64]
        for name in ("amso_delivered", "amso_pending"):
            getattr(lib, name).restype = C.c_int64
            getattr(lib, name).argtypes = [vp]
        lib.amso_generation.restype = C.c_int32
        lib.amso_generation.argtypes = [vp]
        lib.amso_generation_sizes.restype = C.c_int32
        lib.amso_generation_sizes.argtypes = [vp, vp, C.c_int32]
        lib.amso_get_assignment.argtypes = [vp, vp, vp]
        lib.amso_get_messages.argtypes = [vp] + [vp] * 8
        lib.amso_eval_cost.argtypes = [vp, vp, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        lib.amso_update_table.argtypes = [vp, C.c_int32, vp]
        lib.amso_destroy.argtypes = [vp]
        _LIBS[dtype] = lib
    return _LIBS[dtype]


class OracleAMaxSum:
    """Same surface as pydcop_amd.amaxsum.AMaxSumEngine."""

    def __init__(self, graph: FlatGraph, params: Params = None):
        self.graph = graph
        self.params = params or Params()
        self._lib = _lib(self.params.dtype)
        cg, cp = graph.to_c(), self.params.to_c()
        self._h = self._lib.amso_create(C.byref(cg), C.byref(cp))

    def reset(self):
        self._lib.amso_reset(self._h)

    def update_factor_table(self, factor: int, table):
        t = np.ascontiguousarray(table, dtype=np.float64).reshape(-1)
        lo, hi = int(self.graph.table_off[factor]), int(self.graph.table_off[factor + 1])
        assert t.shape[0] == hi - lo
        self._lib.amso_update_table(self._h, int(factor), t.ctypes.data)

    def run(self, max_generations: int = -1, max_messages: int = -1) -> int:
        """Deliver the queued messages of generations < max_generations (all: -1)."""
        return int(self._lib.amso_run(self._h, int(max_generations), int(max_messages)))

    @property
    def delivered(self) -> int:
        return int(self._lib.amso_delivered(self._h))

    @property
    def pending(self) -> int:
        return int(self._lib.amso_pending(self._h))

    @property
    def generation(self) -> int:
        return int(self._lib.amso_generation(self._h))

    def generation_sizes(self):
        buf = np.zeros(1 << 16, dtype=np.int64)
        n = self._lib.amso_generation_sizes(self._h, buf.ctypes.data, buf.shape[0])
        return buf[:n].copy()

    def assignment(self):
        idx = np.empty(self.graph.n_vars, dtype=np.int32)
        belief = np.empty(self.graph.n_vars, dtype=np.float64)
        self._lib.amso_get_assignment(self._h, idx.ctypes.data, belief.ctypes.data)
        return idx, belief

    def messages(self) -> dict:
        nm, ne = int(self.graph.msg_off[-1]), self.graph.n_edges
        out = {k: np.empty(nm) for k in ("f_cost", "v_cost", "f_prev", "v_prev")}
        out.update({k: np.empty(ne, dtype=np.uint8) for k in ("f_has", "v_has", "f_cnt", "v_cnt")})
        self._lib.amso_get_messages(self._h, *[out[k].ctypes.data for k in
                                               ("f_cost", "v_cost", "f_prev", "v_prev", "f_has", "v_has", "f_cnt", "v_cnt")])
        return out

    def eval_cost(self, idx=None, infinity=float("inf")):
        cost, viol = C.c_double(), C.c_int64()
        p = None
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            p = idx.ctypes.data
        self._lib.amso_eval_cost(self._h, p, float(infinity), C.byref(cost), C.byref(viol))
        return cost.value, int(viol.value)

    def close(self):
        if self._h:
            self._lib.amso_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
