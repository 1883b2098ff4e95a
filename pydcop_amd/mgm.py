"""MGM (pydcop/algorithms/mgm.py) on the GPU: the ctypes binding of the `mxs_mgm_*` entry points
(include/maxsum_gpu.h; device code: pydcop_amd/csrc/mgm.hip) on the same FlatGraph as the Max-Sum
engine -- factors are the constraints, variables the MGM computations.  No CPU fallback."""
import ctypes as C
from typing import Optional, Tuple

import numpy as np

from .engine import MaxSumGpuError, load_library
from .graph import FlatGraph, Params


def name_ranks(names) -> np.ndarray:
    """Rank of every variable's name in sorted order: MGM breaks ties with `sorted(names)`
    (mgm.py:566-575)."""
    order = sorted(range(len(names)), key=lambda i: names[i])
    rank = np.empty(len(names), dtype=np.int32)
    rank[order] = np.arange(len(names), dtype=np.int32)
    return rank


class MgmEngine:
    """>>> eng = MgmEngine(graph, Params(mode="min"))   # every computation started
    >>> eng.run(30)                                    # 30 rounds (= the reference's stop_cycle 31)
    >>> idx, cost = eng.assignment()
    """

    def __init__(self, graph: FlatGraph, params: Optional[Params] = None, device: int = 0,
                 lib_path: Optional[str] = None):
        self._h = None
        self._lib = load_library(lib_path)
        self.graph = graph
        self.params = params or Params()
        cg, cp = graph.to_c(), self.params.to_c()
        self._rank = name_ranks(graph.var_names) if graph.var_names else None
        h = C.c_void_p()
        self._check(self._lib.mxs_mgm_create(C.byref(cg), C.byref(cp),
                                             None if self._rank is None else self._rank.ctypes.data,
                                             int(device), C.byref(h)))
        self._h = h
        # cost ties of a variable without neighbours break on the domain VALUE (relations.py:1661-1665):
        # only needed when some domain is not written in ascending order
        self._vrank = graph.value_rank()
        if self._vrank is not None:
            self._check(self._lib.mxs_mgm_set_value_rank(self._h, self._vrank.ctypes.data))

    def _check(self, rc: int):
        if rc != 0:
            raise MaxSumGpuError(f"maxsum_gpu error {rc}: {self._lib.mxs_last_error().decode()}")

    def reset(self):
        self._check(self._lib.mxs_mgm_reset(self._h))

    def run(self, n_rounds: int):
        self._check(self._lib.mxs_mgm_run(self._h, int(n_rounds)))

    @property
    def cycle_count(self) -> int:
        n = C.c_int64(0)
        self._check(self._lib.mxs_mgm_rounds(self._h, C.byref(n)))
        return int(n.value)

    def state(self) -> dict:
        n = self.graph.n_vars
        out = {"idx": np.empty(n, dtype=np.int32), "cost": np.empty(n), "has_cost": np.empty(n, dtype=np.uint8),
               "gain": np.empty(n), "new": np.empty(n, dtype=np.int32)}
        self._check(self._lib.mxs_mgm_get_state(self._h, *[out[k].ctypes.data for k in
                                                           ("idx", "cost", "has_cost", "gain", "new")]))
        return out

    def assignment(self) -> Tuple[np.ndarray, np.ndarray]:
        s = self.state()
        return s["idx"], s["cost"]

    def eval_cost(self, idx=None, infinity: float = float("inf")) -> Tuple[float, int]:
        cost, viol = C.c_double(0), C.c_int64(0)
        p = None
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            p = idx.ctypes.data
        self._check(self._lib.mxs_mgm_eval_cost(self._h, p, float(infinity), C.byref(cost), C.byref(viol)))
        return float(cost.value), int(viol.value)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mxs_mgm_destroy(self._h)
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
