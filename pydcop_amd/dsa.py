"""DSA (pydcop/algorithms/dsa.py, variants A / B / C) on the GPU: the ctypes binding of the
`mxs_dsa_*` entry points (include/maxsum_gpu.h; device code: pydcop_amd/csrc/dsa.hip) on the same
FlatGraph as the Max-Sum engine.  Every stochastic choice comes from a counter-based generator keyed
on (seed, variable, cycle, draw): a run is reproducible and independent of scheduling.  No CPU
fallback."""
import ctypes as C
from typing import Optional, Tuple

import numpy as np

from .engine import MaxSumGpuError, load_library
from .graph import FlatGraph, Params

VARIANTS = {"A": 0, "B": 1, "C": 2}


class DsaEngine:
    """>>> eng = DsaEngine(graph, Params(mode="min"), variant="B", probability=0.7, seed=1)
    >>> eng.run(30)                                    # 30 cycles (= the reference's stop_cycle 30)
    >>> idx, cost = eng.assignment()
    """

    def __init__(self, graph: FlatGraph, params: Optional[Params] = None, variant: str = "B",
                 probability: float = 0.7, p_mode: str = "fixed", seed: int = 0, device: int = 0,
                 lib_path: Optional[str] = None):
        if variant not in VARIANTS:
            raise ValueError(f"Invalid value {variant!r} for parameter variant, must be one of ['A', 'B', 'C']")
        if p_mode not in ("fixed", "arity"):
            raise ValueError(f"Invalid value {p_mode!r} for parameter p_mode, must be one of ['arity', 'fixed']")
        self._h = None
        self._lib = load_library(lib_path)
        self.graph = graph
        self.params = params or Params()
        cg, cp = graph.to_c(), self.params.to_c()
        h = C.c_void_p()
        self._check(self._lib.mxs_dsa_create(C.byref(cg), C.byref(cp), VARIANTS[variant], float(probability),
                                             1 if p_mode == "arity" else 0, int(seed) & (2 ** 64 - 1),
                                             int(device), C.byref(h)))
        self._h = h
        # cost ties of a variable without neighbours break on the domain VALUE (relations.py:1661-1665):
        # only needed when some domain is not written in ascending order
        self._vrank = graph.value_rank()
        if self._vrank is not None:
            self._check(self._lib.mxs_dsa_set_value_rank(self._h, self._vrank.ctypes.data))

    def _check(self, rc: int):
        if rc != 0:
            raise MaxSumGpuError(f"maxsum_gpu error {rc}: {self._lib.mxs_last_error().decode()}")

    def reset(self):
        self._check(self._lib.mxs_dsa_reset(self._h))

    def run(self, n_cycles: int):
        self._check(self._lib.mxs_dsa_run(self._h, int(n_cycles)))

    @property
    def cycle_count(self) -> int:
        n = C.c_int64(0)
        self._check(self._lib.mxs_dsa_cycles(self._h, C.byref(n)))
        return int(n.value)

    def assignment(self) -> Tuple[np.ndarray, np.ndarray]:
        idx = np.empty(self.graph.n_vars, dtype=np.int32)
        cost = np.empty(self.graph.n_vars)
        self._check(self._lib.mxs_dsa_get_state(self._h, idx.ctypes.data, cost.ctypes.data))
        return idx, cost

    def eval_cost(self, idx=None, infinity: float = float("inf")) -> Tuple[float, int]:
        cost, viol = C.c_double(0), C.c_int64(0)
        p = None
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            p = idx.ctypes.data
        self._check(self._lib.mxs_dsa_eval_cost(self._h, p, float(infinity), C.byref(cost), C.byref(viol)))
        return float(cost.value), int(viol.value)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mxs_dsa_destroy(self._h)
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
