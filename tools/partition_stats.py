"""Partition statistics of a BASELINE configuration (no GPU needed): how much crosses between the
shards when `pydcop_amd.partition.partition_variables` (csrc/partition.cpp, the stand-in for the
METIS north_star names) cuts the instance k ways -- cut-factor fraction, halo elements per rank and
per peer, imbalance, wall time.  profiles/partition_<workload>_k<k>.json

usage: python tools/partition_stats.py [workload] [k ...]
       python tools/partition_stats.py weak [k ...]     # the N > 1 default of bench.py: ONE instance of
                                                        # k x 100k variables of the metric's family, cut k ways"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bench import make_workload  # noqa: E402
from pydcop_amd.partition import build_shard, cut_statistics, partition_variables  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stats(workload, k, scale=1):
    g, _ = make_workload(workload, scale)
    t0 = time.perf_counter()
    part = partition_variables(g, k)
    t1 = time.perf_counter()
    st = cut_statistics(g, part)
    shards = [build_shard(g, part, r, k) for r in range(k)]
    t2 = time.perf_counter()
    send = np.array([[int(x) for x in sh.send_counts] for sh in shards])   # elements (message entries) rank -> peer
    D = int(g.dom_size.max())
    owned = np.array([sh.n_owned for sh in shards])
    out = {"workload": workload + (f" x{scale} (weak scaling: {g.n_vars // scale} variables per rank)" if scale > 1 else ""), "k": k, "n_vars": g.n_vars, "n_factors": g.n_factors, "n_edges": g.n_edges,
           "partitioner": "pydcop_amd/csrc/partition.cpp (multilevel: heavy-edge matching, greedy growing, "
                          "boundary FM; METIS is not installed)",
           "partition_wall_s": round(t1 - t0, 2), "build_shards_wall_s": round(t2 - t1, 2),
           "cut_factor_fraction": st["cut_fraction"], "edge_imbalance": st["edge_imbalance"],
           "owned_vars_per_rank": {"min": int(owned.min()), "max": int(owned.max())},
           "factors_per_rank_incl_replicas": {"min": int(min(sh.graph.n_factors for sh in shards)),
                                              "max": int(max(sh.graph.n_factors for sh in shards))},
           "halo_elements_sent_per_rank": {"min": int(send.sum(1).min()), "max": int(send.sum(1).max()),
                                           "total": int(send.sum())},
           "halo_records_sent_per_rank_max": int(send.sum(1).max()) // D,
           "halo_elements_per_peer_pair": {"min": int(send[send > 0].min()) if (send > 0).any() else 0,
                                           "max": int(send.max()), "pairs_with_traffic": int((send > 0).sum())},
           "halo_bytes_per_rank_per_cycle": {"f64": int(send.sum(1).max()) * 8, "f32": int(send.sum(1).max()) * 4},
           "note": "a cut factor is replicated on every shard owning one of its variables; only V->F messages "
                   "of cut edges cross, once per cycle (DESIGN section 6)"}
    return out


if __name__ == "__main__":
    workload = sys.argv[1] if len(sys.argv) > 1 else "coloring_1m_deg6"
    ks = [int(x) for x in sys.argv[2:]] or [2, 4, 8]
    weak = workload == "weak"
    for k in ks:
        rec = stats("coloring_100k", k, k) if weak else stats(workload, k)
        path = os.path.join(ROOT, "profiles", f"partition_coloring_100k_x{k}_k{k}.json" if weak else f"partition_{workload}_k{k}.json")
        with open(path, "w") as f:
            json.dump(rec, f, indent=1)
        print(json.dumps(rec))
