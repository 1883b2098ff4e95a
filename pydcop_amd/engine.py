"""ctypes binding of the C-ABI in include/maxsum_gpu.h (libmaxsum_hip.so).

`MaxSumEngine` is the array-level entry point of the batched GPU sweep that
replaces the reference's per-agent message loop
(pydcop/algorithms/maxsum.py:279-565 driven by
pydcop/infrastructure/computations.py:633-829).  There is no CPU fallback: if the
HIP library is missing or no MI355X is visible, construction raises.
"""
import ctypes as C
import os
import sys
from typing import Optional, Tuple

import numpy as np

from .graph import CGraph, CParams, FlatGraph, Params

_HERE = os.path.dirname(os.path.abspath(__file__))
# $MAXSUM_HIP_LIB selects another HIP build of the same library (kernel A/B experiments:
# `make -C pydcop_amd/csrc variant NAME=x` -> libmaxsum_hip_x.so).  It cannot select anything
# that is not a hipcc build: load_library checks mxs_build_kind().
DEFAULT_LIB = os.environ.get("MAXSUM_HIP_LIB") or os.path.join(_HERE, "csrc", "libmaxsum_hip.so")
_TEST_ENGINES = set()


def register_test_engine(path: str, make_default: bool = False):
    """TESTS ONLY (tests/conftest.py, tests/emu/run_emulated.py): allow the host emulation of
    the engine sources (tests/emu, mxs_build_kind() == 0) to be loaded in this process.  There
    is deliberately no environment variable or parameter that does this: the product path
    cannot be pointed at a CPU build from outside."""
    global DEFAULT_LIB
    _TEST_ENGINES.add(os.path.realpath(path))
    if make_default:
        DEFAULT_LIB = os.path.realpath(path)

# every symbol include/maxsum_gpu.h declares
ABI_SYMBOLS = (
    "mxs_device_count", "mxs_create", "mxs_reset", "mxs_run", "mxs_run_timed", "mxs_run_reps",
    "mxs_run_async", "mxs_sync", "mxs_cycle_count", "mxs_get_assignment",
    "mxs_get_messages", "mxs_eval_cost", "mxs_cycle_bytes", "mxs_halo_setup",
    "mxs_halo_buffers", "mxs_halo_bind", "mxs_step_compute", "mxs_step_pack", "mxs_step_unpack", "mxs_stream",
    "mxs_comm_unique_id", "mxs_comm_init", "mxs_comm_exchange", "mxs_run_sharded", "mxs_shard_mode",
    "mxs_peer_export", "mxs_peer_connect",
    "mxs_debug_timeline", "mxs_update_factor_table", "mxs_destroy", "mxs_last_error", "mxs_version",
    "mxs_build_kind", "mxs_set_state", "mxs_set_parent_table", "mxs_slice_factor",
    "mxs_table_storage", "mxs_factor_order", "mxs_factor_kernels", "mxs_variable_kernels",
    "mxs_amaxsum_create", "mxs_amaxsum_reset", "mxs_amaxsum_run", "mxs_amaxsum_status",
    "mxs_amaxsum_generation_sizes", "mxs_amaxsum_get_assignment", "mxs_amaxsum_get_messages",
    "mxs_amaxsum_eval_cost", "mxs_amaxsum_update_factor_table", "mxs_amaxsum_destroy",
    "mxs_mgm_create", "mxs_mgm_reset", "mxs_mgm_set_value_rank", "mxs_mgm_run", "mxs_mgm_rounds", "mxs_mgm_get_state",
    "mxs_mgm_eval_cost", "mxs_mgm_destroy",
    "mxs_dsa_create", "mxs_dsa_reset", "mxs_dsa_set_value_rank", "mxs_dsa_run", "mxs_dsa_cycles", "mxs_dsa_get_state",
    "mxs_dsa_eval_cost", "mxs_dsa_destroy",
)


MAX_PEERS = 8


class PeerInfo(C.Structure):
    """struct mxs_peer_info (include/maxsum_gpu.h): what a rank tells the others for the
    peer-store exchange.  Plain bytes: pickles through torch.distributed."""
    _fields_ = [("qualifies", C.c_int32), ("rank", C.c_int32), ("ghost_len", C.c_int64),
                ("recv_at", C.c_int64 * MAX_PEERS), ("recv_len", C.c_int64 * MAX_PEERS),
                ("ghost_handle", C.c_uint8 * 64), ("flag_handle", C.c_uint8 * 64),
                ("pid", C.c_int64), ("ghost_ptr", C.c_uint64), ("flag_ptr", C.c_uint64)]


class MaxSumGpuError(RuntimeError):
    """A C-ABI call failed (the message comes from mxs_last_error)."""


_libs = {}
_hip_runtime = None


def hip_runtime_path() -> str:
    """The HIP runtime (libamdhip64) this process uses.

    libmaxsum_hip.so is linked without it so that there is exactly ONE runtime per
    process: torch bundles its own copy, and the multi-GPU path hands the engine's
    stream and halo buffers to torch.distributed (RCCL), which only works inside
    one runtime.  Order: $MAXSUM_HIP_RUNTIME, the copy bundled with an already
    imported torch, then the system ROCm."""
    env = os.environ.get("MAXSUM_HIP_RUNTIME")
    if env:
        return env
    torch = sys.modules.get("torch")
    if torch is not None:
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            return cand
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        cand = os.path.join(rocm, "lib", name)
        if os.path.exists(cand):
            return cand
    return "libamdhip64.so"


def _load_hip_runtime():
    global _hip_runtime
    if _hip_runtime is None:
        path = hip_runtime_path()
        try:
            _hip_runtime = (path, C.CDLL(path, mode=C.RTLD_GLOBAL))
        except OSError as e:
            raise MaxSumGpuError(f"cannot load the HIP runtime {path}: {e}. "
                                 "maxsum_gpu has no CPU fallback.") from e
    return _hip_runtime[0]


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Load the engine library and declare the prototypes of its C-ABI."""
    path = os.path.realpath(path or DEFAULT_LIB)  # (symlinks and relative paths name the same library)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise MaxSumGpuError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). maxsum_gpu has no CPU fallback.")
    registered_test = path in _TEST_ENGINES
    if not registered_test:
        if not os.path.basename(path).startswith("libmaxsum_hip"):
            raise MaxSumGpuError(
                f"{path} is not a build of libmaxsum_hip: maxsum_gpu has no CPU fallback and loads "
                "only the gfx950 library (pydcop_amd/csrc/libmaxsum_hip*.so)")
        _load_hip_runtime()  # (the emulated test build carries its own fake runtime)
    lib = C.CDLL(path)
    lib.mxs_build_kind.restype = C.c_int32
    if lib.mxs_build_kind() != 1 and not registered_test:
        raise MaxSumGpuError(f"{path} is not a hipcc/gfx950 build (mxs_build_kind() == "
                             f"{lib.mxs_build_kind()}): maxsum_gpu has no CPU fallback")
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    protos = {
        "mxs_device_count": ([C.POINTER(i32)], C.c_int),
        "mxs_create": ([C.POINTER(CGraph), C.POINTER(CParams), i32, C.POINTER(vp)], C.c_int),
        "mxs_reset": ([vp], C.c_int),
        "mxs_run": ([vp, i32], C.c_int),
        "mxs_run_timed": ([vp, i32, C.POINTER(C.c_float)], C.c_int),
        "mxs_run_reps": ([vp, i32, i32, vp], C.c_int),
        "mxs_run_async": ([vp, i32], C.c_int),
        "mxs_sync": ([vp], C.c_int),
        "mxs_cycle_count": ([vp, C.POINTER(i64)], C.c_int),
        "mxs_get_assignment": ([vp, vp, vp], C.c_int),
        "mxs_get_messages": ([vp, vp, vp, vp, vp], C.c_int),
        "mxs_eval_cost": ([vp, vp, C.c_double, C.POINTER(C.c_double), C.POINTER(i64)], C.c_int),
        "mxs_set_state": ([vp, vp, vp, vp, vp, vp, vp, i64], C.c_int),
        "mxs_set_parent_table": ([vp, i32, vp, i32, vp, vp], C.c_int),
        "mxs_slice_factor": ([vp, i32, vp], C.c_int),
        "mxs_table_storage": ([vp, vp, C.POINTER(i64)], C.c_int),
        "mxs_amaxsum_create": ([C.POINTER(CGraph), C.POINTER(CParams), i32, C.POINTER(vp)], C.c_int),
        "mxs_amaxsum_reset": ([vp], C.c_int),
        "mxs_amaxsum_update_factor_table": ([vp, i32, vp, i64], C.c_int),
        "mxs_amaxsum_run": ([vp, i32, C.POINTER(i64)], C.c_int),
        "mxs_amaxsum_status": ([vp, C.POINTER(i32), C.POINTER(i64), C.POINTER(i64)], C.c_int),
        "mxs_amaxsum_generation_sizes": ([vp, vp, i32, C.POINTER(i32)], C.c_int),
        "mxs_amaxsum_get_assignment": ([vp, vp, vp], C.c_int),
        "mxs_amaxsum_get_messages": ([vp] + [vp] * 8, C.c_int),
        "mxs_amaxsum_eval_cost": ([vp, vp, C.c_double, C.POINTER(C.c_double), C.POINTER(i64)], C.c_int),
        "mxs_amaxsum_destroy": ([vp], C.c_int),
        "mxs_mgm_create": ([C.POINTER(CGraph), C.POINTER(CParams), vp, i32, C.POINTER(vp)], C.c_int),
        "mxs_mgm_reset": ([vp], C.c_int),
        "mxs_mgm_set_value_rank": ([vp, vp], C.c_int),
        "mxs_mgm_run": ([vp, i32], C.c_int),
        "mxs_mgm_rounds": ([vp, C.POINTER(i64)], C.c_int),
        "mxs_mgm_get_state": ([vp, vp, vp, vp, vp, vp], C.c_int),
        "mxs_mgm_eval_cost": ([vp, vp, C.c_double, C.POINTER(C.c_double), C.POINTER(i64)], C.c_int),
        "mxs_mgm_destroy": ([vp], C.c_int),
        "mxs_dsa_create": ([C.POINTER(CGraph), C.POINTER(CParams), i32, C.c_double, i32, C.c_uint64, i32,
                            C.POINTER(vp)], C.c_int),
        "mxs_dsa_reset": ([vp], C.c_int),
        "mxs_dsa_set_value_rank": ([vp, vp], C.c_int),
        "mxs_dsa_run": ([vp, i32], C.c_int),
        "mxs_dsa_cycles": ([vp, C.POINTER(i64)], C.c_int),
        "mxs_dsa_get_state": ([vp, vp, vp], C.c_int),
        "mxs_dsa_eval_cost": ([vp, vp, C.c_double, C.POINTER(C.c_double), C.POINTER(i64)], C.c_int),
        "mxs_dsa_destroy": ([vp], C.c_int),
        "mxs_cycle_bytes": ([vp, C.POINTER(i64), C.POINTER(i32)], C.c_int),
        "mxs_factor_order": ([vp, C.POINTER(i32)], C.c_int),
        "mxs_factor_kernels": ([vp, vp], C.c_int),
        "mxs_variable_kernels": ([vp, vp], C.c_int),
        "mxs_halo_setup": ([vp, vp, i64, vp, i64], C.c_int),
        "mxs_halo_buffers": ([vp, C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), C.POINTER(i64)], C.c_int),
        "mxs_halo_bind": ([vp, vp, vp], C.c_int),
        "mxs_step_compute": ([vp], C.c_int),
        "mxs_step_pack": ([vp], C.c_int),
        "mxs_step_unpack": ([vp], C.c_int),
        "mxs_stream": ([vp, C.POINTER(vp)], C.c_int),
        "mxs_comm_unique_id": ([C.c_char_p, vp], C.c_int),
        "mxs_comm_init": ([vp, C.c_char_p, i32, i32, vp, vp, vp], C.c_int),
        "mxs_comm_exchange": ([vp], C.c_int),
        "mxs_run_sharded": ([vp, i32], C.c_int),
        "mxs_shard_mode": ([vp, C.POINTER(i32), C.POINTER(i32)], C.c_int),
        "mxs_peer_export": ([vp, i32, i32, vp, vp, C.POINTER(PeerInfo)], C.c_int),
        "mxs_peer_connect": ([vp, C.POINTER(PeerInfo)], C.c_int),
        "mxs_debug_timeline": ([vp, vp, i32, C.POINTER(i32)], C.c_int),
        "mxs_update_factor_table": ([vp, i32, vp, i64], C.c_int),
        "mxs_destroy": ([vp], C.c_int),
        "mxs_last_error": ([], C.c_char_p),
        "mxs_version": ([], i32),
        "mxs_build_kind": ([], i32),
    }
    for name, (argtypes, restype) in protos.items():
        fn = getattr(lib, name)  # AttributeError if the library misses a symbol
        fn.argtypes = argtypes
        fn.restype = restype
    _libs[path] = lib
    return lib


def rccl_path() -> str:
    """The RCCL library this process uses for the native exchange of the sharded path:
    $MAXSUM_RCCL_LIB, the copy bundled with an already imported torch (one copy per
    process, like the HIP runtime), then the system ROCm."""
    env = os.environ.get("MAXSUM_RCCL_LIB")
    if env:
        return env
    torch = sys.modules.get("torch")
    if torch is not None:
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(cand):
            return cand
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    for name in ("librccl.so", "librccl.so.1"):
        cand = os.path.join(rocm, "lib", name)
        if os.path.exists(cand):
            return cand
    return "librccl.so"


UNIQUE_ID_BYTES = 128


def comm_unique_id(lib_path: Optional[str] = None, rccl: Optional[str] = None) -> bytes:
    """ncclGetUniqueId (rank 0 calls it and hands the bytes to the other ranks)."""
    lib = load_library(lib_path)
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    rc = lib.mxs_comm_unique_id((rccl or rccl_path()).encode(), C.cast(buf, C.c_void_p))
    if rc != 0:
        raise MaxSumGpuError(f"maxsum_gpu error {rc}: {lib.mxs_last_error().decode()}")
    return buf.raw


def peer_qualifies(info: bytes) -> bool:
    return bool(PeerInfo.from_buffer_copy(info).qualifies)


def device_count(lib_path: Optional[str] = None) -> int:
    n = C.c_int32(0)
    load_library(lib_path).mxs_device_count(C.byref(n))
    return int(n.value)


class MaxSumEngine:
    """One engine = one GPU sweeping one (shard of a) factor graph.

    >>> eng = MaxSumEngine(graph, Params(mode="min"))     # cycle 0 (start) done
    >>> eng.run(30)                                       # 30 synchronous cycles
    >>> idx, belief = eng.assignment()
    """

    def __init__(self, graph: FlatGraph, params: Optional[Params] = None, device: int = 0,
                 lib_path: Optional[str] = None):
        self._h = None
        self._lib = load_library(lib_path)
        self.graph = graph
        self.params = params or Params()
        cg, cp = graph.to_c(), self.params.to_c()
        h = C.c_void_p()
        self._check(self._lib.mxs_create(C.byref(cg), C.byref(cp), int(device), C.byref(h)))
        self._h = h

    def _check(self, rc: int):
        if rc != 0:
            raise MaxSumGpuError(f"maxsum_gpu error {rc}: {self._lib.mxs_last_error().decode()}")

    # -- cycles --------------------------------------------------------------
    def reset(self):
        self._check(self._lib.mxs_reset(self._h))

    def run(self, n_cycles: int):
        self._check(self._lib.mxs_run(self._h, int(n_cycles)))

    def run_timed(self, n_cycles: int) -> float:
        """Run and return the device time in milliseconds (HIP events on the
        engine's stream)."""
        ms = C.c_float(0)
        self._check(self._lib.mxs_run_timed(self._h, int(n_cycles), C.byref(ms)))
        return float(ms.value)

    def run_reps(self, n_cycles: int, reps: int) -> np.ndarray:
        """`reps` repetitions of `n_cycles` cycles enqueued back to back (one HIP event between two
        repetitions, one host wait at the end) -> the device time of every repetition in
        milliseconds (mxs_run_reps)."""
        ms = np.zeros(int(reps), dtype=np.float32)
        self._check(self._lib.mxs_run_reps(self._h, int(n_cycles), int(reps), ms.ctypes.data))
        return ms.astype(np.float64)

    def run_async(self, n_cycles: int):
        self._check(self._lib.mxs_run_async(self._h, int(n_cycles)))

    def sync(self):
        self._check(self._lib.mxs_sync(self._h))

    @property
    def cycle_count(self) -> int:
        n = C.c_int64(0)
        self._check(self._lib.mxs_cycle_count(self._h, C.byref(n)))
        return int(n.value)

    # -- results ---------------------------------------------------------------
    def assignment(self) -> Tuple[np.ndarray, np.ndarray]:
        idx = np.empty(self.graph.n_vars, dtype=np.int32)
        belief = np.empty(self.graph.n_vars, dtype=np.float64)
        self._check(self._lib.mxs_get_assignment(self._h, idx.ctypes.data, belief.ctypes.data))
        return idx, belief

    def messages(self):
        nm, ne = int(self.graph.msg_off[-1]), self.graph.n_edges
        v2f, f2v = np.empty(nm), np.empty(nm)
        cv, cf = np.empty(ne, dtype=np.uint8), np.empty(ne, dtype=np.uint8)
        self._check(self._lib.mxs_get_messages(self._h, v2f.ctypes.data, f2v.ctypes.data,
                                               cv.ctypes.data, cf.ctypes.data))
        return v2f, f2v, cv, cf

    def state(self) -> dict:
        """Everything a run needs to carry on elsewhere: messages, send counters, selection,
        beliefs (caller's orders) and the cycle count -- `set_state` of another engine on the
        same graph resumes it bit for bit (checkpoint / resume)."""
        v2f, f2v, cv, cf = self.messages()
        idx, belief = self.assignment()
        return {"v2f": v2f, "f2v": f2v, "count_v2f": cv, "count_f2v": cf, "idx": idx, "belief": belief,
                "cycles": self.cycle_count}

    def set_state(self, v2f=None, f2v=None, count_v2f=None, count_f2v=None, idx=None, belief=None,
                  cycles: Optional[int] = None):
        """The inverse of `messages()` / `assignment()` (mxs_set_state); None = keep."""
        nm, ne, nv = int(self.graph.msg_off[-1]), self.graph.n_edges, self.graph.n_vars

        def arr(a, dtype, n, what):
            if a is None:
                return None, None
            a = np.ascontiguousarray(a, dtype=dtype)
            if a.shape != (n,):
                raise ValueError(f"{what} must have {n} entries")
            return a, a.ctypes.data
        keep = []
        ptrs = []
        for a, dt, n, what in ((v2f, np.float64, nm, "v2f"), (f2v, np.float64, nm, "f2v"),
                               (count_v2f, np.uint8, ne, "count_v2f"), (count_f2v, np.uint8, ne, "count_f2v"),
                               (idx, np.int32, nv, "idx"), (belief, np.float64, nv, "belief")):
            a, p = arr(a, dt, n, what)
            keep.append(a)
            ptrs.append(p)
        cyc = self.cycle_count if cycles is None else int(cycles)
        self._check(self._lib.mxs_set_state(self._h, *ptrs, cyc))

    def eval_cost(self, idx=None, infinity: float = float("inf")) -> Tuple[float, int]:
        """(cost, violations) as DCOP.solution_cost (pydcop/dcop/dcop.py:319-367)."""
        cost, viol = C.c_double(0), C.c_int64(0)
        p = None
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            if idx.shape[0] != self.graph.n_vars:
                raise ValueError("assignment must have one index per variable")
            p = idx.ctypes.data
        self._check(self._lib.mxs_eval_cost(self._h, p, float(infinity), C.byref(cost), C.byref(viol)))
        return float(cost.value), int(viol.value)

    def cycle_bytes(self) -> Tuple[int, int]:
        """(algorithmic bytes per cycle, kernel launches per cycle)."""
        b, n = C.c_int64(0), C.c_int32(0)
        self._check(self._lib.mxs_cycle_bytes(self._h, C.byref(b), C.byref(n)))
        return int(b.value), int(n.value)

    def factor_order(self) -> str:
        """"tiled" / "by_first_variable": how the binary factors are ordered inside their classes
        (mxs_factor_order; a layout decision, every order computes the same messages)."""
        t = C.c_int32(0)
        self._check(self._lib.mxs_factor_order(self._h, C.byref(t)))
        return "tiled" if t.value else "by_first_variable"

    def factor_kernels(self) -> dict:
        """Factors per kernel of the factor side (mxs_factor_kernels): register classes, generic
        (thread per edge), workgroup per factor, one wave per factor (box records), lane grid."""
        n = (C.c_int64 * 7)()
        self._check(self._lib.mxs_factor_kernels(self._h, n))
        return dict(zip(("reg_unary", "reg_binary", "generic", "workgroup", "wave_box", "lane_grid", "lane_group_small"),
                        (int(x) for x in n)))

    def variable_kernels(self) -> dict:
        """Variables per kernel of the variable side (mxs_variable_kernels)."""
        n = (C.c_int64 * 6)()
        self._check(self._lib.mxs_variable_kernels(self._h, n))
        return dict(zip(("packed", "packed8", "wide", "generic", "not_swept", "hub"), (int(x) for x in n)))

    def table_storage(self) -> dict:
        """{"full", "f32", "i16", "i8"}: factors per table storage type, and "bytes_per_cycle": the
        table bytes one cycle reads (mxs_table_storage; narrow types are lossless)."""
        n = (C.c_int64 * 4)()
        b = C.c_int64(0)
        self._check(self._lib.mxs_table_storage(self._h, n, C.byref(b)))
        return {"full": int(n[0]), "f32": int(n[1]), "i16": int(n[2]), "i8": int(n[3]),
                "bytes_per_cycle": int(b.value)}

    def update_factor_table(self, factor: int, table):
        """New cost table (same shape, row-major over the scope) for one factor; the
        iteration carries on (maxsum_dynamic.py:80-104, change_factor_function)."""
        t = np.ascontiguousarray(table, dtype=np.float64).reshape(-1)
        self._check(self._lib.mxs_update_factor_table(self._h, int(factor), t.ctypes.data, t.shape[0]))
        lo, hi = int(self.graph.table_off[factor]), int(self.graph.table_off[factor + 1])
        self.graph.tables[lo:hi] = t  # keep the host copy of the graph in step

    def set_parent_table(self, factor: int, parent, is_external):
        """Register the whole relation of a factor that also depends on external (read-only)
        variables: `parent` = ndarray over ALL its dimensions, `is_external[i]` true for the
        read-only ones; the others, in order, are the factor's scope (maxsum_dynamic.py:113-186)."""
        parent = np.ascontiguousarray(parent, dtype=np.float64)
        dims = np.ascontiguousarray(parent.shape, dtype=np.int32)
        ext = np.ascontiguousarray(is_external, dtype=np.uint8)
        if ext.shape != (parent.ndim,):
            raise ValueError("one is_external flag per dimension of the parent relation")
        self._check(self._lib.mxs_set_parent_table(self._h, int(factor), parent.ctypes.data, parent.ndim,
                                                   dims.ctypes.data, ext.ctypes.data))
        self._parents = getattr(self, "_parents", {})
        self._parents[int(factor)] = (parent, ext.astype(bool))

    def slice_factor(self, factor: int, external_idx):
        """The factor's active table becomes the slice of its parent relation at these value
        indices of the external dimensions (one per external dimension, in order); sliced on
        the device.  The iteration carries on."""
        idx = np.ascontiguousarray(external_idx, dtype=np.int32)
        self._check(self._lib.mxs_slice_factor(self._h, int(factor), idx.ctypes.data))
        parent, ext = self._parents[int(factor)]   # keep the host copy of the graph in step
        sel, j = [], 0
        for i in range(parent.ndim):
            if ext[i]:
                sel.append(int(idx[j]))
                j += 1
            else:
                sel.append(slice(None))
        lo, hi = int(self.graph.table_off[factor]), int(self.graph.table_off[factor + 1])
        self.graph.tables[lo:hi] = parent[tuple(sel)].reshape(-1)

    def debug_timeline(self) -> np.ndarray:
        """Profiling: run one more cycle with per-block timestamps; returns an int64
        array [n_blocks, 3] = (start tick, end tick, class kind), 100 MHz ticks."""
        n = C.c_int32(0)
        self._check(self._lib.mxs_debug_timeline(self._h, None, 0, C.byref(n)))
        out = np.zeros((int(n.value), 3), dtype=np.int64)
        self._check(self._lib.mxs_debug_timeline(self._h, out.ctypes.data, int(n.value), C.byref(n)))
        return out

    # -- sharded operation ---------------------------------------------------------
    def halo_setup(self, send_edges, recv_edges):
        s = np.ascontiguousarray(send_edges, dtype=np.int32)
        r = np.ascontiguousarray(recv_edges, dtype=np.int32)
        self._check(self._lib.mxs_halo_setup(self._h, s.ctypes.data, s.shape[0],
                                             r.ctypes.data, r.shape[0]))

    def halo_buffers(self):
        """((send device pointer, bytes), (recv device pointer, bytes))"""
        s, r = C.c_void_p(), C.c_void_p()
        sb, rb = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.mxs_halo_buffers(self._h, C.byref(s), C.byref(sb), C.byref(r), C.byref(rb)))
        return (s.value, int(sb.value)), (r.value, int(rb.value))

    def halo_bind(self, send_ptr: int, recv_ptr: int):
        """Pack into / unpack from caller-owned device buffers (the tensors the
        collective runs on)."""
        self._check(self._lib.mxs_halo_bind(self._h, C.c_void_p(send_ptr), C.c_void_p(recv_ptr)))

    def step_compute(self):
        self._check(self._lib.mxs_step_compute(self._h))

    def step_pack(self):
        self._check(self._lib.mxs_step_pack(self._h))

    def step_unpack(self):
        self._check(self._lib.mxs_step_unpack(self._h))

    def comm_init(self, rank: int, world: int, unique_id: bytes, send_counts, recv_counts,
                  rccl: Optional[str] = None):
        """Join the RCCL communicator of the sharded run (collective over all ranks);
        counts in ELEMENTS per peer, as given by `Shard.send_counts / recv_counts`."""
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        sc = np.ascontiguousarray(send_counts, dtype=np.int64)
        rc = np.ascontiguousarray(recv_counts, dtype=np.int64)
        if sc.shape != (world,) or rc.shape != (world,):
            raise ValueError("one count per rank")
        uid = C.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
        self._check(self._lib.mxs_comm_init(self._h, (rccl or rccl_path()).encode(), int(rank), int(world),
                                            C.cast(uid, C.c_void_p), sc.ctypes.data, rc.ctypes.data))

    def comm_exchange(self):
        self._check(self._lib.mxs_comm_exchange(self._h))

    def run_sharded(self, n_cycles: int):
        """n sharded cycles (compute, RCCL exchange, unpack) enqueued by the library."""
        self._check(self._lib.mxs_run_sharded(self._h, int(n_cycles)))

    def shard_mode(self) -> dict:
        """{"fused_launch", "direct_exchange", "peer_stores"} -- how this shard runs its cycles."""
        f, d = C.c_int32(0), C.c_int32(0)
        self._check(self._lib.mxs_shard_mode(self._h, C.byref(f), C.byref(d)))
        return {"fused_launch": bool(f.value), "direct_exchange": d.value == 1, "peer_stores": d.value == 2}

    def peer_export(self, rank: int, world: int, send_counts, recv_counts) -> bytes:
        """Peer-store exchange, step 1 (after halo_setup): this rank's `mxs_peer_info` as bytes;
        `peer_qualifies(info)` tells whether the shard can run in that mode."""
        sc = np.ascontiguousarray(send_counts, dtype=np.int64)
        rc = np.ascontiguousarray(recv_counts, dtype=np.int64)
        if sc.shape != (world,) or rc.shape != (world,):
            raise ValueError("one count per rank")
        info = PeerInfo()
        self._check(self._lib.mxs_peer_export(self._h, int(rank), int(world), sc.ctypes.data, rc.ctypes.data,
                                              C.byref(info)))
        return bytes(info)

    def peer_connect(self, infos):
        """Step 2: `infos[q]` = the bytes rank q got from `peer_export` (all qualifying)."""
        arr = (PeerInfo * len(infos))()
        for i, b in enumerate(infos):
            C.memmove(C.byref(arr[i]), bytes(b), C.sizeof(PeerInfo))
        self._check(self._lib.mxs_peer_connect(self._h, arr))

    def stream(self) -> int:
        s = C.c_void_p()
        self._check(self._lib.mxs_stream(self._h, C.byref(s)))
        return s.value or 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mxs_destroy(self._h)
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
