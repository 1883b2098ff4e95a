"""Asynchronous Max-Sum (pydcop/algorithms/amaxsum.py) on the GPU, under FIFO delivery, one
generation of messages per step -- the ctypes binding of the `mxs_amaxsum_*` entry points
(include/maxsum_gpu.h; device code: pydcop_amd/csrc/amaxsum.hip).  No CPU fallback."""
import ctypes as C
from typing import Optional, Tuple

import numpy as np

from .engine import MaxSumGpuError, load_library
from .graph import FlatGraph, Params


class AMaxSumEngine:
    """>>> eng = AMaxSumEngine(graph, Params(start_messages="leafs_vars"))   # every computation started
    >>> eng.run()                 # until no message is left (or run(max_generations=20))
    >>> idx, cost = eng.assignment()
    """

    def __init__(self, graph: FlatGraph, params: Optional[Params] = None, device: int = 0,
                 lib_path: Optional[str] = None):
        self._h = None
        self._lib = load_library(lib_path)
        self.graph = graph
        self.params = params or Params()
        cg, cp = graph.to_c(), self.params.to_c()
        h = C.c_void_p()
        self._check(self._lib.mxs_amaxsum_create(C.byref(cg), C.byref(cp), int(device), C.byref(h)))
        self._h = h

    def _check(self, rc: int):
        if rc != 0:
            raise MaxSumGpuError(f"maxsum_gpu error {rc}: {self._lib.mxs_last_error().decode()}")

    def reset(self):
        self._check(self._lib.mxs_amaxsum_reset(self._h))

    def run(self, max_generations: int = -1) -> int:
        """Deliver whole generations while the next one's number is < max_generations (-1: until
        the queue is empty); returns the number of messages handled."""
        n = C.c_int64(0)
        self._check(self._lib.mxs_amaxsum_run(self._h, int(max_generations), C.byref(n)))
        return int(n.value)

    def _status(self):
        g, p, d = C.c_int32(0), C.c_int64(0), C.c_int64(0)
        self._check(self._lib.mxs_amaxsum_status(self._h, C.byref(g), C.byref(p), C.byref(d)))
        return int(g.value), int(p.value), int(d.value)

    @property
    def generation(self) -> int:
        """Number of the last generation handled (-1 before the first)."""
        return self._status()[0] - 1

    @property
    def pending(self) -> int:
        return self._status()[1]

    @property
    def delivered(self) -> int:
        return self._status()[2]

    def generation_sizes(self) -> np.ndarray:
        n = C.c_int32(0)
        self._check(self._lib.mxs_amaxsum_generation_sizes(self._h, None, 0, C.byref(n)))
        out = np.zeros(max(int(n.value), 1), dtype=np.int64)
        self._check(self._lib.mxs_amaxsum_generation_sizes(self._h, out.ctypes.data, out.shape[0], C.byref(n)))
        return out[:int(n.value)]

    def assignment(self) -> Tuple[np.ndarray, np.ndarray]:
        idx = np.empty(self.graph.n_vars, dtype=np.int32)
        belief = np.empty(self.graph.n_vars, dtype=np.float64)
        self._check(self._lib.mxs_amaxsum_get_assignment(self._h, idx.ctypes.data, belief.ctypes.data))
        return idx, belief

    def messages(self) -> dict:
        nm, ne = int(self.graph.msg_off[-1]), self.graph.n_edges
        out = {k: np.empty(nm) for k in ("f_cost", "v_cost", "f_prev", "v_prev")}
        out.update({k: np.empty(ne, dtype=np.uint8) for k in ("f_has", "v_has", "f_cnt", "v_cnt")})
        self._check(self._lib.mxs_amaxsum_get_messages(
            self._h, *[out[k].ctypes.data for k in ("f_cost", "v_cost", "f_prev", "v_prev",
                                                    "f_has", "v_has", "f_cnt", "v_cnt")]))
        return out

    def update_factor_table(self, factor: int, table):
        """`change_factor_function` with the same scope (pydcop/algorithms/maxsum_dynamic.py:80-104):
        a new table (in the factor's own dimension order) between two generations."""
        t = np.ascontiguousarray(table, dtype=np.float64).reshape(-1)
        self._check(self._lib.mxs_amaxsum_update_factor_table(self._h, int(factor), t.ctypes.data, t.shape[0]))
        lo, hi = int(self.graph.table_off[factor]), int(self.graph.table_off[factor + 1])
        if t.shape[0] == hi - lo:
            self.graph.tables[lo:hi] = t

    def eval_cost(self, idx=None, infinity: float = float("inf")) -> Tuple[float, int]:
        cost, viol = C.c_double(0), C.c_int64(0)
        p = None
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            p = idx.ctypes.data
        self._check(self._lib.mxs_amaxsum_eval_cost(self._h, p, float(infinity), C.byref(cost), C.byref(viol)))
        return float(cost.value), int(viol.value)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mxs_amaxsum_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
