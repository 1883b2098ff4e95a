"""Max-Sum on a factor graph partitioned across the GPUs of one node: one process
per GPU, one `MaxSumEngine` per process, one exchange of boundary V->F messages
per synchronous cycle.

The reference's counterpart is message passing between agents
(pydcop/infrastructure/communication.py:588-698); here the only messages that
leave a GPU are those of cut factors' remote variables:

    cycle t:   compute stream   variables + interior factors of cycle t, then -- once
                                the exchange of cycle t-1 is unpacked -- the cut factors
               comm stream      pack (after the variables of cycle t) -> all-to-all
                                (RCCL over xGMI) -> unpack into the ghost slots

Three exchanges (`collective=`; default "auto" = "rccl" on GPUs -- the RCCL all-to-all that
BASELINE.json's north_star names -- and "torch" for host-memory engines):
"p2p" (OPT-IN: `collective="p2p"` or MAXSUM_COLLECTIVE=p2p; fastest in the one-GPU loopback
measurements but never yet run between two physical GPUs, so not the default) -- no collective
at all: the ranks of one node map each other's ghost buffers through
hipIpc, the variable kernel stores cut-edge records straight into the peer's ghost region
(xGMI peer stores), and a cycle is ONE launch whose cut factor blocks poll flag words the
peers publish (mxs_peer_export / mxs_peer_connect; shards with binary cut factors only);
"rccl" and "torch" -- an all-to-all, one group of ncclSend / ncclRecv per peer with the fixed
counts of the partition, between the two launches of a cycle as drawn above.  On GPUs the engine calls RCCL itself (`collective="rccl"`: the whole
cycle loop stays inside libmaxsum_hip.so, mxs_run_sharded -- no interpreter between
two 50-microsecond cycles; torch.distributed only bootstraps the communicator and
gathers results); `collective="torch"` runs the same exchange as
torch.distributed.all_to_all_single on the engine's comm stream (any backend; what the
gloo CPU tests use, and the fallback when RCCL cannot be loaded).

Both message directions of cycle t read only cycle t-1 (Jacobi schedule), so one
exchange per cycle is enough, a suppressed (unsent) message needs no special case
(the sender's record simply still holds the old value), and -- because only the cut
factors of cycle t+1 read what the exchange of cycle t delivers -- the exchange has
a whole cycle of other work to hide behind.
"""
import os
import warnings
from typing import Optional, Tuple

import numpy as np

from .engine import MaxSumEngine, MaxSumGpuError, comm_unique_id, peer_qualifies
from .graph import FlatGraph, Params
from .partition import Shard, build_shard, partition_variables


class ShardedMaxSum:
    """Same surface as `MaxSumEngine`, over `world` ranks.

    Every rank passes the same (whole) graph and partition; a rank only uploads
    its shard.  `torch.distributed` must be initialised (nccl on GPUs; gloo works
    for host-memory engines, which is what the CPU tests use)."""

    def __init__(self, graph: FlatGraph, params: Optional[Params], rank: int, world: int,
                 device: int = 0, part: Optional[np.ndarray] = None,
                 lib_path: Optional[str] = None, group=None, collective: str = "auto",
                 rccl: Optional[str] = None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.rank, self.world, self.group = rank, world, group
        self.graph = graph
        self.params = params or Params()
        self.part = partition_variables(graph, world) if part is None else np.asarray(part, dtype=np.int32)
        self.shard: Shard = build_shard(graph, self.part, rank, world)
        self._device, self._lib_path = device, lib_path
        self.engine = MaxSumEngine(self.shard.graph, self.params, device=device, lib_path=lib_path)
        self.engine.halo_setup(self.shard.send_edges, self.shard.recv_edges)
        backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self._backend = backend
        self._on_gpu = backend == "nccl"
        collective = os.environ.get("MAXSUM_COLLECTIVE") or collective
        if collective == "auto":
            # GPUs: the engine's own RCCL all-to-all (north_star's exchange); host-memory engines
            # (tests): torch's all_to_all.  Peer stores are opt-in until a run on two or more
            # physical GPUs has shown bit-parity with the single engine (bench.py checks it).
            collective = "rccl" if self._on_gpu else "torch"
        if collective not in ("p2p", "rccl", "torch"):
            raise ValueError("collective must be 'auto', 'p2p', 'rccl' or 'torch'")
        self._p2p = collective == "p2p" and self._init_p2p()
        if self._p2p:
            self._native = True  # the cycle loop is in the library here too
            self.collective = "p2p"
            self.engine.sync()
            return
        if collective == "p2p":
            collective = "rccl" if self._on_gpu else "torch"
        self._native = collective == "rccl" and self._init_native(lib_path, rccl)
        self.collective = "rccl" if self._native else "torch"
        if self._native:
            self.engine.comm_exchange()  # the start messages of cycle 0
            self.engine.step_unpack()
            self.engine.sync()
            return
        tdtype = torch.float64 if self.params.dtype == "f64" else torch.float32
        tdev = torch.device("cuda", device) if self._on_gpu else torch.device("cpu")
        n_send, n_recv = int(self.shard.send_counts.sum()), int(self.shard.recv_counts.sum())
        # the collective runs on tensors torch owns; the engine packs into / unpacks
        # from them directly (mxs_halo_bind)
        self._send = torch.zeros(max(n_send, 1), dtype=tdtype, device=tdev)
        self._recv = torch.zeros(max(n_recv, 1), dtype=tdtype, device=tdev)
        self._n_send, self._n_recv = n_send, n_recv
        self._send_splits = [int(x) for x in self.shard.send_counts]
        self._recv_splits = [int(x) for x in self.shard.recv_counts]
        self._stream_ctx = None
        if self._on_gpu:
            # torch's collectives order themselves against the *current* stream:
            # make that the engine's comm stream
            self._ext_stream = torch.cuda.ExternalStream(self.engine.stream(), device=tdev)
        if self._on_gpu:
            torch.cuda.synchronize(tdev)
        self.engine.halo_bind(self._send.data_ptr(), self._recv.data_ptr())
        self._exchange()          # the start messages of cycle 0
        self.engine.step_unpack()
        self.engine.sync()

    def _init_p2p(self) -> bool:
        """Peer-store exchange (no collective): every rank describes its ghost buffers, and only
        if ALL shards qualify do they map each other's memory (hipIpc over xGMI)."""
        dist = self._dist
        if self.world < 2 or self._backend == "none":
            return False
        try:
            info = self.engine.peer_export(self.rank, self.world, self.shard.send_counts, self.shard.recv_counts)
        except MaxSumGpuError as e:
            warnings.warn(f"peer-store exchange unavailable: {e}")
            info = None
        infos = [None] * self.world
        dist.all_gather_object(infos, info, group=self.group)
        if any(i is None or not peer_qualifies(i) for i in infos):
            if info is not None and peer_qualifies(info):
                self._fresh_engine()  # drop what peer_export has set up on this rank
            return False
        err = None
        try:
            self.engine.peer_connect(infos)
        except MaxSumGpuError as e:  # e.g. hipIpcOpenMemHandle refused
            err = str(e)
        oks = [None] * self.world
        dist.all_gather_object(oks, err is None, group=self.group)  # also: every rank has pushed
        if not all(oks):
            # a connected engine cannot go back: start over with a fresh one for the collective
            warnings.warn(f"peer-store exchange unavailable ({err or 'another rank failed'}); using RCCL")
            self._fresh_engine()
            return False
        return True

    def _fresh_engine(self):
        self.engine.close()
        self.engine = MaxSumEngine(self.shard.graph, self.params, device=self._device, lib_path=self._lib_path)
        self.engine.halo_setup(self.shard.send_edges, self.shard.recv_edges)

    def _init_native(self, lib_path, rccl) -> bool:
        """Create the engine's own RCCL communicator.  Every rank first checks that it can
        load RCCL, and only if ALL can do they enter ncclCommInitRank (a rank that failed
        on its own would leave the others waiting inside it)."""
        dist = self._dist
        multi = self.world > 1 and self._backend != "none"
        uid, err = None, None
        try:
            uid = comm_unique_id(lib_path, rccl)  # rank 0's is the one used
        except MaxSumGpuError as e:
            err = str(e)
        if multi:
            oks = [None] * self.world
            dist.all_gather_object(oks, err is None, group=self.group)
            box = [uid]
            dist.broadcast_object_list(box, src=0, group=self.group)
            uid = box[0]
            ok = all(oks)
        else:
            ok = err is None
        if not ok:
            warnings.warn(f"native RCCL exchange unavailable ({err or 'another rank failed'}); "
                          "falling back to torch.distributed.all_to_all_single")
            return False
        err = None
        try:
            self.engine.comm_init(self.rank, self.world, uid, self.shard.send_counts,
                                  self.shard.recv_counts, rccl=rccl)
        except MaxSumGpuError as e:  # e.g. ncclCommInitRank refused on this node
            err = str(e)
        if multi:
            oks = [None] * self.world
            dist.all_gather_object(oks, err is None, group=self.group)
            ok = all(oks)
        else:
            ok = err is None
        if not ok:
            # an engine that joined (or half-joined) a communicator cannot go back
            warnings.warn(f"native RCCL exchange unavailable ({err or 'another rank failed'}); "
                          "falling back to torch.distributed.all_to_all_single")
            self._fresh_engine()
            return False
        return True

    # -- the per-cycle exchange ------------------------------------------------------
    def _exchange(self):
        if self._native:
            self.engine.comm_exchange()
            return
        if self._backend == "none":
            return  # a single engine without torch.distributed: nothing crosses
        torch, dist = self._torch, self._dist
        send = self._send[:self._n_send]
        recv = self._recv[:self._n_recv]
        if self.world == 1:
            # nothing crosses either, but keep the collective in the stream (one padding
            # element to itself): the single-GPU test then runs the same RCCL code path
            send, recv = self._send[:1], self._recv[:1]
            splits = ([1], [1])
        else:
            splits = (self._recv_splits, self._send_splits)
        if self._on_gpu:
            with torch.cuda.stream(self._ext_stream):
                dist.all_to_all_single(recv, send, splits[0], splits[1], group=self.group)
        else:
            self.engine.sync()
            dist.all_to_all_single(recv, send, splits[0], splits[1], group=self.group)

    def run_async(self, n_cycles: int):
        if self._native:
            self.engine.run_sharded(n_cycles)  # the loop is in the library
            return
        step, unpack = self.engine.step_compute, self.engine.step_unpack  # compute + pack
        if self._on_gpu and self.world > 1:
            # the host loop is on the critical path of short cycles: enter the comm-stream
            # context once and call the collective with pre-bound arguments
            torch, dist = self._torch, self._dist
            send, recv = self._send[:self._n_send], self._recv[:self._n_recv]
            rs, ss, group = self._recv_splits, self._send_splits, self.group
            with torch.cuda.stream(self._ext_stream):
                for _ in range(int(n_cycles)):
                    step()
                    dist.all_to_all_single(recv, send, rs, ss, group=group)
                    unpack()
            return
        for _ in range(int(n_cycles)):
            step()
            self._exchange()
            unpack()

    def sync(self):
        self.engine.sync()

    def run(self, n_cycles: int):
        self.run_async(n_cycles)
        self.sync()

    def reset(self):
        if self._p2p:
            self._dist.barrier(group=self.group)  # no peer still runs cycles of the previous run
            self.engine.reset()
            return
        self.engine.reset()
        self._exchange()
        self.engine.step_unpack()
        self.engine.sync()

    @property
    def cycle_count(self) -> int:
        return self.engine.cycle_count

    # -- results -----------------------------------------------------------------------
    def _all_gather_owned(self, values: np.ndarray) -> np.ndarray:
        """Owned entries of every rank -> one global array (same on all ranks)."""
        out = np.empty(self.graph.n_vars, dtype=values.dtype)
        sh = self.shard
        if self.world == 1:
            out[sh.local_vars[:sh.n_owned]] = values[:sh.n_owned]
            return out
        parts = [None] * self.world
        self._dist.all_gather_object(parts, (sh.local_vars[:sh.n_owned], values[:sh.n_owned]),
                                     group=self.group)
        for ids, vals in parts:
            out[ids] = vals
        return out

    def assignment(self) -> Tuple[np.ndarray, np.ndarray]:
        """(idx, belief) of the WHOLE graph, gathered from the owners."""
        idx, belief = self.engine.assignment()
        return self._all_gather_owned(idx), self._all_gather_owned(belief)

    def eval_cost(self, idx=None, infinity: float = float("inf")) -> Tuple[float, int]:
        """DCOP.solution_cost (pydcop/dcop/dcop.py:319-367) of a whole-graph
        assignment (default: the current selection): every shard evaluates the
        factors and variables it owns, then one all-reduce."""
        if idx is None:
            idx = self.assignment()[0]
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        cost, viol = self.engine.eval_cost(idx[self.shard.local_vars], infinity)
        if self.world == 1:
            return cost, viol
        parts = [None] * self.world
        self._dist.all_gather_object(parts, (cost, viol), group=self.group)
        return float(sum(p[0] for p in parts)), int(sum(p[1] for p in parts))

    def close(self):
        self.engine.close()


_LEAKED = []   # engines whose rank thread never came back from a C-ABI call (see _each)


class LocalShardedMaxSum:
    """The sharded path driven from ONE process: k shards on k GPUs, one thread per GPU.

    This is what the plugin's `devices` parameter and `api.solve_*(devices=k)` use -- pyDCOP's
    thread mode hosts every computation in one process, so the ranks of
    the exchange are threads of that process, each stepping its own engine through the
    library's cycle loop (`mxs_run_sharded`) over the engine-owned RCCL communicator
    (`mxs_comm_init`; one communicator rank per device, created concurrently from the k threads
    as RCCL requires).  ctypes releases the GIL inside every C-ABI call, so the k ranks really
    run side by side.  Same surface as `MaxSumEngine`; same arithmetic as one engine sweeping
    the whole graph (a shard inherits the global orders, `partition.build_shard`)."""

    def __init__(self, graph: FlatGraph, params: Optional[Params], devices, part: Optional[np.ndarray] = None,
                 lib_path: Optional[str] = None, rccl: Optional[str] = None):
        self.graph = graph
        self.params = params or Params()
        self.devices = [int(d) for d in devices]
        k = self.world = len(self.devices)
        if k < 1:
            raise ValueError("at least one device")
        if len(set(self.devices)) != k:
            raise ValueError("devices must be distinct (RCCL refuses two ranks on one GPU)")
        self.part = partition_variables(graph, k) if part is None else np.asarray(part, dtype=np.int32)
        self.shards = [build_shard(graph, self.part, r, k) for r in range(k)]
        self.engines = [None] * k
        self._stuck = set()
        self.collective = "rccl" if k > 1 else "none"
        uid = comm_unique_id(lib_path, rccl) if k > 1 else None

        # Two phases with a join in between: ncclCommInitRank blocks until ALL k ranks have
        # called it, so a rank must not enter it before every rank is known to get there (an
        # engine that failed to build -- out of memory, bad device, bad halo lists -- would
        # leave the other k - 1 threads waiting for it forever, with the plug-in's session lock
        # held).  The multi-process driver all-gathers an ok flag for the same reason.
        def make(r):
            s = self.shards[r]
            e = MaxSumEngine(s.graph, self.params, device=self.devices[r], lib_path=lib_path)
            self.engines[r] = e
            if k > 1:
                e.halo_setup(s.send_edges, s.recv_edges)
            e.sync()

        def connect(r):
            s, e = self.shards[r], self.engines[r]
            e.comm_init(r, k, uid, s.send_counts, s.recv_counts, rccl=rccl)
            e.comm_exchange()  # the start messages of cycle 0
            e.step_unpack()
            e.sync()
        try:
            self._each(make)
            if k > 1:
                self._each(connect, timeout=self.COMM_TIMEOUT_S)
        except Exception:
            self.close()
            raise

    # how long the ranks may take to meet in the communicator before the caller gets an error
    # instead of a hang (a wedged rank's thread is a daemon: it does not keep the process alive)
    COMM_TIMEOUT_S = 300.0

    def _each(self, fn, timeout=None):
        """fn(rank) on one thread per rank; the first exception is re-raised here.  With a
        `timeout` (seconds, all ranks together) a rank that has not returned raises
        MaxSumGpuError in the caller instead of blocking it."""
        if self.world == 1:
            return [fn(0)]
        import threading
        import time
        out, errors = [None] * self.world, []

        def work(r):
            try:
                out[r] = fn(r)
            except Exception as e:  # surfaced in the calling thread
                errors.append((r, e))
        threads = [threading.Thread(target=work, args=(r,), name=f"maxsum-gpu-rank{r}", daemon=True)
                   for r in range(self.world)]
        for t in threads:
            t.start()
        deadline = None if timeout is None else time.monotonic() + timeout
        for t in threads:
            t.join(None if deadline is None else max(0.0, deadline - time.monotonic()))
        stuck = [r for r, t in enumerate(threads) if t.is_alive()]
        if stuck:
            # their threads are still inside a C-ABI call on their engines: those handles are
            # leaked on purpose (close() skips them) -- destroying one under a running call
            # would be a use after free
            self._stuck.update(stuck)
            first = f"; rank {errors[0][0]} failed first: {errors[0][1]}" if errors else ""
            raise MaxSumGpuError(f"ranks {stuck} (devices {[self.devices[r] for r in stuck]}) did not return within "
                                 f"{timeout:.0f} s{first}")
        if errors:
            r, e = errors[0]
            raise MaxSumGpuError(f"rank {r} (device {self.devices[r]}): {e}") from e
        return out

    def run(self, n_cycles: int):
        n = int(n_cycles)

        def go(r):
            e = self.engines[r]
            if self.world > 1:
                e.run_sharded(n)
                e.sync()
            else:
                e.run(n)
        self._each(go)

    def sync(self):
        self._each(lambda r: self.engines[r].sync())

    def reset(self):
        def go(r):
            e = self.engines[r]
            e.reset()
            if self.world > 1:
                e.comm_exchange()
                e.step_unpack()
            e.sync()
        self._each(go)

    @property
    def cycle_count(self) -> int:
        return self.engines[0].cycle_count

    def assignment(self) -> Tuple[np.ndarray, np.ndarray]:
        idx = np.empty(self.graph.n_vars, dtype=np.int32)
        belief = np.empty(self.graph.n_vars, dtype=np.float64)
        for s, e in zip(self.shards, self.engines):
            i, b = e.assignment()
            idx[s.local_vars[:s.n_owned]] = i[:s.n_owned]
            belief[s.local_vars[:s.n_owned]] = b[:s.n_owned]
        return idx, belief

    def eval_cost(self, idx=None, infinity: float = float("inf")) -> Tuple[float, int]:
        if idx is None:
            idx = self.assignment()[0]
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        parts = [e.eval_cost(idx[s.local_vars], infinity) for s, e in zip(self.shards, self.engines)]
        return float(sum(p[0] for p in parts)), int(sum(p[1] for p in parts))

    def update_factor_table(self, factor: int, table):
        """`MaxSumEngine.update_factor_table` on every shard that holds a replica of the factor."""
        t = np.ascontiguousarray(table, dtype=np.float64).reshape(-1)
        for s, e in zip(self.shards, self.engines):
            pos = np.searchsorted(s.local_factors, factor)
            if pos < s.local_factors.shape[0] and s.local_factors[pos] == factor:
                e.update_factor_table(int(pos), t)
        lo, hi = int(self.graph.table_off[factor]), int(self.graph.table_off[factor + 1])
        self.graph.tables[lo:hi] = t

    def close(self):
        for r, e in enumerate(self.engines):
            if e is None:
                continue
            if r in self._stuck:
                _LEAKED.append(e)   # keeps __del__ from destroying it under the running call
            else:
                e.close()
        self.engines = [None] * self.world

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
