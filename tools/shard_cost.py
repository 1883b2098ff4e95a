"""What one rank of the weak-scaling bench computes per cycle (profiling, one GPU):
build the N x 100k-variable instance, partition it N-way, create the engine of shard 0
and step it WITHOUT the collective (pack / unpack still run).  Prints cut statistics,
halo volume and the per-cycle time of the shard.
usage: python tools/shard_cost.py [N] [dtype]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import make_workload
from pydcop_amd.engine import MaxSumEngine
from pydcop_amd.graph import Params
from pydcop_amd.partition import build_shard, cut_statistics, partition_variables

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dt = sys.argv[2] if len(sys.argv) > 2 else "f64"
g, mode = make_workload("coloring_100k", N)
t0 = time.perf_counter()
part = partition_variables(g, N)
t1 = time.perf_counter()
st = cut_statistics(g, part)
sh = build_shard(g, part, 0, N)
t2 = time.perf_counter()
e = MaxSumEngine(sh.graph, Params(mode=mode, dtype=dt))
e.halo_setup(sh.send_edges, sh.recv_edges)
def cycles(n):
    for _ in range(n):
        e.step_compute(); e.step_unpack()
    e.sync()
cycles(200)
t3 = time.perf_counter(); cycles(2000); t4 = time.perf_counter()
word = 8 if dt == "f64" else 4
print(json.dumps({"ranks": N, "n_vars": g.n_vars, "partition_s": t1 - t0, "build_shard_s": t2 - t1,
                  "cut_fraction": st["cut_fraction"], "edge_imbalance": st["edge_imbalance"],
                  "shard_vars_owned": sh.n_owned, "shard_vars_ghost": int(sh.graph.n_vars - sh.n_owned),
                  "shard_factors": sh.graph.n_factors,
                  "halo_send_bytes": int(sh.send_counts.sum()) * word, "halo_recv_bytes": int(sh.recv_counts.sum()) * word,
                  "shard_cycle_us_without_collective": 1e6 * (t4 - t3) / 2000}))
