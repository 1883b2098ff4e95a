"""Test helpers for the sharded path (TEST INFRASTRUCTURE).

`LocalShards` runs the k shards of a partition as k engines in ONE process and
moves the packed halo buffers between them with tensor copies -- everything of
pydcop_amd.sharded except the collective itself.  On the CPU it drives the
emulated-engine build, on a GPU box k engines on the same MI355X.

Run as a script it is one rank of the gloo / nccl test:
    python tests/shard_harness.py <world> <rank> <port> <lib or -> <out.npz> <case> <cycles...>
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pydcop_amd.engine import MaxSumEngine  # noqa: E402
from pydcop_amd.graph import Params  # noqa: E402
from pydcop_amd.partition import build_shard, partition_variables  # noqa: E402


def make_case(name):
    from pydcop_amd import generators as G
    if name == "coloring":
        return G.random_coloring(600, avg_degree=4, seed=5, names=False), {}
    if name == "mixed_max":
        return G.random_mixed(150, 260, seed=3, names=False), {"mode": "max", "start_messages": "all"}
    if name == "ising":
        return G.ising_grid(16, 12, seed=2, names=False), {"start_messages": "leafs_vars"}
    if name == "coloring_deg9":
        return G.random_coloring(300, avg_degree=9, seed=8, names=False), {"damping_nodes": "vars"}
    if name == "coloring_2k":
        return G.random_coloring(2000, avg_degree=4, seed=11, names=False), {}
    if name == "scalefree":      # hub variables (the wave-per-edges class, round 6) on the boundary and inside
        return G.scalefree_coloring(1500, m=3, seed=21, names=False), {}
    if name == "secp":           # small-domain lane-group kernel + arity-5 factors cut across shards
        return G.secp_like(120, 80, 100, max_model_size=4, seed=22, names=False), {"start_messages": "all"}
    if name == "multi":          # arity-6 models (the workgroup kernel in passes, round 6) cut across shards
        return G.secp_like(70, 50, 50, max_model_size=5, seed=23, names=False), {}
    if name == "coloring_50k":
        return G.random_coloring(50_000, avg_degree=4, seed=1, names=False), {}
    raise ValueError(name)


class LocalShards:
    def __init__(self, graph, params, k, lib_path=None, device="cpu", part=None):
        import torch
        self.torch = torch
        self.graph, self.k = graph, k
        self.part = partition_variables(graph, k) if part is None else part
        self.shards = [build_shard(graph, self.part, r, k) for r in range(k)]
        self.engines = [MaxSumEngine(s.graph, params, lib_path=lib_path) for s in self.shards]
        tdt = torch.float64 if params.dtype == "f64" else torch.float32
        self.send, self.recv = [], []
        for s, e in zip(self.shards, self.engines):
            e.halo_setup(s.send_edges, s.recv_edges)
            self.send.append(torch.zeros(max(int(s.send_counts.sum()), 1), dtype=tdt, device=device))
            self.recv.append(torch.zeros(max(int(s.recv_counts.sum()), 1), dtype=tdt, device=device))
        if device != "cpu":
            torch.cuda.synchronize()
        for e, a, b in zip(self.engines, self.send, self.recv):
            e.halo_bind(a.data_ptr(), b.data_ptr())
        self._exchange()

    def _exchange(self):
        for e in self.engines:
            e.sync()
        for r, s in enumerate(self.shards):       # receiver r
            roff = 0
            for q in range(self.k):               # from q
                n = int(s.recv_counts[q])
                if n:
                    soff = int(self.shards[q].send_counts[:r].sum())
                    assert int(self.shards[q].send_counts[r]) == n
                    self.recv[r][roff:roff + n].copy_(self.send[q][soff:soff + n])
                roff += n
        if self.send[0].device.type != "cpu":
            self.torch.cuda.synchronize()
        for e in self.engines:
            e.step_unpack()
            e.sync()

    def run(self, n):
        for _ in range(n):
            for e in self.engines:
                e.step_compute()   # compute + pack
            self._exchange()

    def assignment(self):
        idx = np.empty(self.graph.n_vars, dtype=np.int32)
        bel = np.empty(self.graph.n_vars)
        for s, e in zip(self.shards, self.engines):
            i, b = e.assignment()
            idx[s.local_vars[:s.n_owned]] = i[:s.n_owned]
            bel[s.local_vars[:s.n_owned]] = b[:s.n_owned]
        return idx, bel

    def eval_cost(self, infinity=float("inf")):
        idx = self.assignment()[0]
        tot = [e.eval_cost(idx[s.local_vars], infinity) for s, e in zip(self.shards, self.engines)]
        return sum(t[0] for t in tot), sum(t[1] for t in tot)

    def close(self):
        for e in self.engines:
            e.close()


def rank_main(argv):
    world, rank, port = int(argv[0]), int(argv[1]), int(argv[2])
    lib = None if argv[3] == "-" else argv[3]
    if lib:  # the emulated engine of the CPU tests
        from pydcop_amd import engine
        engine.register_test_engine(lib)
    out, case = argv[4], argv[5]
    steps = [int(x) for x in argv[6:]]
    import torch
    import torch.distributed as dist
    from pydcop_amd.sharded import ShardedMaxSum
    # MAXSUM_TEST_BACKEND=gloo: two ranks on ONE GPU (RCCL refuses that; the peer-store
    # exchange does not need a collective, only a bootstrap)
    backend = os.environ.get("MAXSUM_TEST_BACKEND") or ("gloo" if lib else "nccl")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        torch.cuda.set_device(0)
        # single node: bootstrap over loopback, no InfiniBand probing
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
    dist.init_process_group(backend, rank=rank, world_size=world)
    g, kw = make_case(case)
    run = ShardedMaxSum(g, Params(**kw), rank, world, device=0, lib_path=lib)
    res = {}
    done = 0
    for n in steps:
        run.run(n)
        done += n
        idx, bel = run.assignment()
        cost, viol = run.eval_cost()
        res[f"idx_{done}"], res[f"bel_{done}"] = idx, bel
        res[f"cost_{done}"] = np.array([cost, viol])
    assert run.cycle_count == done
    res["collective"] = np.array(run.collective)
    if rank == 0:
        np.savez(out, **res)
    run.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    rank_main(sys.argv[1:])
