"""ctypes wrapper of oracle/maxsum_oracle.c -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; never by pydcop_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from pydcop_amd.graph import CGraph, CParams, FlatGraph, Params

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force=False):
    """Compile the C restatement (both precisions) with gcc."""
    targets = [os.path.join(_HERE, f"lib{a}_oracle_{p}.so") for a in ("maxsum", "amaxsum", "mgm", "dsa") for p in ("f64", "f32")]
    srcs = [os.path.join(_HERE, "maxsum_oracle.c"), os.path.join(_HERE, "amaxsum_oracle.c"),
            os.path.join(_HERE, "mgm_oracle.c"), os.path.join(_HERE, "dsa_oracle.c")]
    newest = max(os.path.getmtime(s) for s in srcs)
    stale = force or any((not os.path.exists(t)) or os.path.getmtime(t) < newest for t in targets)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return targets


def _lib(dtype):
    if dtype not in _LIBS:
        path = os.path.join(_HERE, f"libmaxsum_oracle_{dtype}.so")
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        lib.mso_create.restype = C.c_void_p
        lib.mso_create.argtypes = [C.POINTER(CGraph), C.POINTER(CParams)]
        lib.mso_reset.argtypes = [C.c_void_p]
        lib.mso_set_threads.argtypes = [C.c_void_p, C.c_int]
        lib.mso_run.argtypes = [C.c_void_p, C.c_int32]
        lib.mso_cycle_count.restype = C.c_int64
        lib.mso_cycle_count.argtypes = [C.c_void_p]
        lib.mso_get_assignment.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.mso_get_messages.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        lib.mso_set_v2f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint8]
        lib.mso_set_state.argtypes = [C.c_void_p] + [C.c_void_p] * 6 + [C.c_int64]
        lib.mso_eval_cost.argtypes = [C.c_void_p, C.c_void_p, C.c_double,
                                      C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        lib.mso_update_table.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        lib.mso_destroy.argtypes = [C.c_void_p]
        _LIBS[dtype] = lib
    return _LIBS[dtype]


class OracleMaxSum:
    """Same surface as pydcop_amd.engine.MaxSumEngine, computed on the CPU."""

    def __init__(self, graph: FlatGraph, params: Params = None, threads: int = 1):
        self.graph = graph
        self.params = params or Params()
        self._lib = _lib(self.params.dtype)
        cg, cp = graph.to_c(), self.params.to_c()
        self._h = self._lib.mso_create(C.byref(cg), C.byref(cp))
        self._lib.mso_set_threads(self._h, threads)

    def reset(self):
        self._lib.mso_reset(self._h)

    def run(self, n_cycles: int):
        self._lib.mso_run(self._h, int(n_cycles))

    @property
    def cycle_count(self) -> int:
        return int(self._lib.mso_cycle_count(self._h))

    def assignment(self):
        idx = np.empty(self.graph.n_vars, dtype=np.int32)
        belief = np.empty(self.graph.n_vars, dtype=np.float64)
        self._lib.mso_get_assignment(self._h, idx.ctypes.data, belief.ctypes.data)
        return idx, belief

    def messages(self):
        nm = int(self.graph.msg_off[-1])
        ne = self.graph.n_edges
        v2f, f2v = np.empty(nm), np.empty(nm)
        cv, cf = np.empty(ne, dtype=np.uint8), np.empty(ne, dtype=np.uint8)
        self._lib.mso_get_messages(self._h, v2f.ctypes.data, f2v.ctypes.data,
                                   cv.ctypes.data, cf.ctypes.data)
        return v2f, f2v, cv, cf

    def state(self) -> dict:
        v2f, f2v, cv, cf = self.messages()
        idx, belief = self.assignment()
        return {"v2f": v2f, "f2v": f2v, "count_v2f": cv, "count_f2v": cf, "idx": idx, "belief": belief,
                "cycles": self.cycle_count}

    def set_state(self, v2f=None, f2v=None, count_v2f=None, count_f2v=None, idx=None, belief=None,
                  cycles=None):
        keep, ptrs = [], []
        for a, dt in ((v2f, np.float64), (f2v, np.float64), (count_v2f, np.uint8), (count_f2v, np.uint8),
                      (idx, np.int32), (belief, np.float64)):
            a = None if a is None else np.ascontiguousarray(a, dtype=dt)
            keep.append(a)
            ptrs.append(None if a is None else a.ctypes.data)
        self._lib.mso_set_state(self._h, *ptrs, self.cycle_count if cycles is None else int(cycles))

    def set_v2f(self, edge: int, msg, cnt: int):
        msg = np.ascontiguousarray(msg, dtype=np.float64)
        self._lib.mso_set_v2f(self._h, int(edge), msg.ctypes.data, int(cnt))

    def update_factor_table(self, factor: int, table):
        t = np.ascontiguousarray(table, dtype=np.float64).reshape(-1)
        self._lib.mso_update_table(self._h, int(factor), t.ctypes.data)

    def eval_cost(self, idx=None, infinity=float("inf")):
        cost, viol = C.c_double(), C.c_int64()
        p = None
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            p = idx.ctypes.data
        self._lib.mso_eval_cost(self._h, p, float(infinity), C.byref(cost), C.byref(viol))
        return cost.value, int(viol.value)

    def close(self):
        if self._h:
            self._lib.mso_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
