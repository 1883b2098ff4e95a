"""A stated PREDICTION of the strong-scaling curve of BASELINE configs[3] (1M-variable degree-6 colouring,
north_star's ">= 6x at 8 GPUs"), from what ONE GPU can measure -- so that the first SCALE run on a multi-GPU
node is a test of numbers written down beforehand:

  * T1: the whole instance on one engine (us per cycle);
  * per N in {2, 4, 8}: the instance partitioned N ways (csrc/partition.cpp), the engine of shard 0 stepped
      (a) without any collective (both launches of a sharded cycle + pack / unpack)  = the shard's compute,
      (b) through the library's own cycle loop with a ONE-rank RCCL communicator that sends the shard's whole
          halo to itself (real ncclSend / ncclRecv of the real volume: the collective's launch + protocol
          latency, not the xGMI hop);
  * the halo volume per rank and cycle, and the number of peers (7 xGMI links of ~153 GB/s per GPU:
    MI355X_MICROARCH.md), give the wire time the loopback cannot show.

predicted cycle(N) = max(compute, exchange) if the exchange hides behind the interior work (it is issued on the
comm stream after launch 1, only the cut factors of the NEXT cycle wait for it), compute + exchange if it does
not; exchange = (loopback cycle - compute, floored at 0) + halo bytes / (links used x 153 GB/s).
usage: python tools/scale_prediction.py [--out profiles/scale_prediction.json]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bench import make_workload  # noqa: E402
from pydcop_amd.engine import MaxSumEngine, comm_unique_id  # noqa: E402
from pydcop_amd.graph import Params  # noqa: E402
from pydcop_amd.partition import build_shard, cut_statistics, partition_variables  # noqa: E402

XGMI_LINK_GBPS = 153.0
STEPS = 600


def timed(fn, sync, steps=STEPS):
    fn(steps // 10)
    sync()
    t = time.perf_counter()
    fn(steps)
    sync()
    return 1e6 * (time.perf_counter() - t) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--workload", default="coloring_1m_deg6")
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--ranks", type=int, nargs="*", default=[2, 4, 8])
    ap.add_argument("--layout-flags", type=int, default=0, help="65536: cut binary factors compute both messages (round 3)")
    a = ap.parse_args()
    g, mode = make_workload(a.workload)
    word = 8 if a.dtype == "f64" else 4
    p = Params(mode=mode, dtype=a.dtype, layout_flags=a.layout_flags)
    with MaxSumEngine(g, p) as e:
        t1 = timed(e.run, e.sync)
    out = {"workload": a.workload, "dtype": a.dtype, "layout_flags": a.layout_flags, "n_vars": g.n_vars, "n_factors": g.n_factors,
           "one_gpu_us_per_cycle": t1, "xgmi_link_gbps": XGMI_LINK_GBPS, "measured_on": "one MI355X (no multi-GPU node)",
           "ranks": []}
    for n in a.ranks:
        part = partition_variables(g, n)
        st = cut_statistics(g, part)
        sh = build_shard(g, part, 0, n)
        n_send, n_recv = int(sh.send_counts.sum()), int(sh.recv_counts.sum())
        peers = int((np.asarray(sh.send_counts) > 0).sum())
        rec = {"n": n, "cut_fraction": st["cut_fraction"], "shard0_owned_vars": int(sh.n_owned),
               "shard0_factors": int(sh.graph.n_factors), "halo_send_bytes": n_send * word, "halo_recv_bytes": n_recv * word,
               "peers_with_traffic": peers}
        e = MaxSumEngine(sh.graph, p)
        e.halo_setup(sh.send_edges, sh.recv_edges)

        def cycles(k):
            for _ in range(k):
                e.step_compute()
                e.step_unpack()
        rec["shard_compute_us"] = timed(cycles, e.sync)
        e.close()
        if n_send == n_recv:
            e = MaxSumEngine(sh.graph, p)
            e.halo_setup(sh.send_edges, sh.recv_edges)
            e.comm_init(0, 1, comm_unique_id(), [n_send], [n_recv])
            rec["shard_mode"] = e.shard_mode()
            rec["shard_cycle_us_rccl_loopback"] = timed(e.run_sharded, e.sync)
            e.close()
        loop = rec.get("shard_cycle_us_rccl_loopback")
        latency = max(0.0, loop - rec["shard_compute_us"]) if loop is not None else None
        wire = rec["halo_send_bytes"] / (max(1, min(peers, 7)) * XGMI_LINK_GBPS * 1e3)  # us
        rec["exchange_latency_us_from_loopback"] = latency
        rec["exchange_wire_us_at_link_rate"] = wire
        if latency is not None:
            ex = latency + wire
            hidden, exposed = max(rec["shard_compute_us"], ex), rec["shard_compute_us"] + ex
            rec["predicted_cycle_us"] = {"exchange_hidden": hidden, "exchange_exposed": exposed}
            rec["predicted_speedup_vs_one_gpu"] = {"exchange_hidden": t1 / hidden, "exchange_exposed": t1 / exposed}
        out["ranks"].append(rec)
        print(json.dumps(rec), flush=True)
    text = json.dumps(out, indent=1)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
