"""What one rank of the weak-scaling bench computes per cycle (profiling, one GPU):
build the N x 100k-variable instance, partition it N-way, create the engine of shard 0
and step it
  (a) WITHOUT any collective (compute + pack + unpack),
  (b) through the library's own cycle loop with a one-rank RCCL communicator that loops
      the shard's whole halo back to itself (mxs_run_sharded: real ncclSend/ncclRecv of
      the real volume, no interpreter in the loop),
  (c) through torch.distributed.all_to_all_single of the same volume on the comm stream
      (one Python iteration per cycle),
so that the cost of the exchange and of the host loop can be told apart on a box with a
single GPU.  Values of (b)/(c) runs are meaningless (the halo is looped back), times are not.
usage: python tools/shard_cost.py [N] [dtype]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from bench import make_workload
from pydcop_amd.engine import MaxSumEngine, comm_unique_id
from pydcop_amd.graph import Params
from pydcop_amd.partition import build_shard, cut_statistics, partition_variables

FLAGS = int(os.environ.get("MAXSUM_LAYOUT_FLAGS", "0"))  # experiments, e.g. 512 = all factors in launch 2
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dt = sys.argv[2] if len(sys.argv) > 2 else "f64"
STEPS = 2000
g, mode = make_workload("coloring_100k", N)
t0 = time.perf_counter()
part = partition_variables(g, N)
t1 = time.perf_counter()
st = cut_statistics(g, part)
sh = build_shard(g, part, 0, N)
t2 = time.perf_counter()
word = 8 if dt == "f64" else 4
n_send, n_recv = int(sh.send_counts.sum()), int(sh.recv_counts.sum())
out = {"ranks": N, "n_vars": g.n_vars, "partition_s": t1 - t0, "build_shard_s": t2 - t1,
       "cut_fraction": st["cut_fraction"], "edge_imbalance": st["edge_imbalance"],
       "shard_vars_owned": sh.n_owned, "shard_vars_ghost": int(sh.graph.n_vars - sh.n_owned),
       "shard_factors": sh.graph.n_factors,
       "halo_send_bytes": n_send * word, "halo_recv_bytes": n_recv * word}


def timed(fn, sync):
    fn(200); sync()
    t = time.perf_counter(); fn(STEPS); sync()
    return 1e6 * (time.perf_counter() - t) / STEPS


ONLY = os.environ.get("MAXSUM_COST_ONLY", "abcd")  # which of the three loops to time
for k in ("MAXSUM_COMM_CUS", "MAXSUM_SHARD_FUSED", "MAXSUM_SHARD_DIRECT", "MAXSUM_LAYOUT_FLAGS"):
    if os.environ.get(k):
        out[k] = os.environ[k]
if "a" not in ONLY:
    pass
# (a) no collective
e = MaxSumEngine(sh.graph, Params(mode=mode, dtype=dt, layout_flags=FLAGS))
e.halo_setup(sh.send_edges, sh.recv_edges)
def cycles_a(n):
    for _ in range(n):
        e.step_compute(); e.step_unpack()
if "a" in ONLY:
    out["shard_cycle_us_without_collective"] = timed(cycles_a, e.sync)
e.close()

# (b) library loop + real RCCL, halo looped back (needs n_send == n_recv)
if n_send == n_recv and "b" in ONLY:
    e = MaxSumEngine(sh.graph, Params(mode=mode, dtype=dt, layout_flags=FLAGS))
    e.halo_setup(sh.send_edges, sh.recv_edges)
    e.comm_init(0, 1, comm_unique_id(), [n_send], [n_recv])
    out["shard_mode"] = e.shard_mode()
    out["shard_cycle_us_native_rccl_loopback"] = timed(e.run_sharded, e.sync)
    e.close()

# (d) peer-store exchange, looped back: the engine is every one of its own peers -- its
# "remote" stores land in its own ghost regions (where peer q's block would be) and its publish
# kernel sets the flag word of every peer.  One fused launch + the publish per cycle, the real
# in-kernel wait on the previous launch's flags; what is missing is the xGMI hop.
if "d" in ONLY and N <= 8:
    from pydcop_amd.engine import PeerInfo, peer_qualifies
    import ctypes as C
    e = MaxSumEngine(sh.graph, Params(mode=mode, dtype=dt, layout_flags=FLAGS))
    e.halo_setup(sh.send_edges, sh.recv_edges)
    mine = e.peer_export(0, N, sh.send_counts, sh.recv_counts)
    out["peer_store_qualifies"] = peer_qualifies(mine)
    if peer_qualifies(mine) and np.array_equal(sh.send_counts, sh.recv_counts):
        me = PeerInfo.from_buffer_copy(mine)
        infos = [mine]
        for q in range(1, N):
            fake = PeerInfo.from_buffer_copy(mine)
            fake.rank = q
            fake.recv_at[0], fake.recv_len[0] = me.recv_at[q], me.recv_len[q]
            fake.flag_ptr = me.flag_ptr + 4 * q   # its word [0] (= me) is my word [q]
            infos.append(bytes(fake))
        e.peer_connect(infos)
        out["shard_mode_peer_stores"] = e.shard_mode()
        out["shard_cycle_us_peer_stores_loopback"] = timed(e.run_sharded, e.sync)
    e.close()

# (c) torch.distributed on the comm stream, same volume
if "c" not in ONLY:
    print(json.dumps(out))
    sys.exit(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo"); os.environ.setdefault("NCCL_IB_DISABLE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
e = MaxSumEngine(sh.graph, Params(mode=mode, dtype=dt, layout_flags=FLAGS))
e.halo_setup(sh.send_edges, sh.recv_edges)
tdt = torch.float64 if dt == "f64" else torch.float32
send = torch.zeros(max(n_send, 1), dtype=tdt, device="cuda")
recv = torch.zeros(max(n_recv, 1), dtype=tdt, device="cuda")
torch.cuda.synchronize()
e.halo_bind(send.data_ptr(), recv.data_ptr())
ext = torch.cuda.ExternalStream(e.stream(), device=torch.device("cuda", 0))
def cycles_c(n):
    with torch.cuda.stream(ext):
        for _ in range(n):
            e.step_compute()
            dist.all_to_all_single(recv, send, [n_recv], [n_send])
            e.step_unpack()
def sync_c():
    e.sync(); torch.cuda.synchronize()
out["shard_cycle_us_torch_all_to_all_loopback"] = timed(cycles_c, sync_c)
e.close()
dist.destroy_process_group()
print(json.dumps(out))
