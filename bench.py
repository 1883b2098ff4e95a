#!/usr/bin/env python
"""bench.py -- Max-Sum iterations/s on the north-star instance (BASELINE.json):
random 3-colouring, 100k variables, average degree 4, binary factors, synchronous
Max-Sum, reference arithmetic (f64).

A "step" is one synchronous Max-Sum cycle over the whole factor graph (every
F->V and V->F message recomputed once + value selection) = one k_sweep launch.
Inputs are resident in HBM before the timed region (the graph is uploaded at
engine creation).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
                    [--dtype f64|f32] [--no-cpu-baseline]

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU):
WEAK scaling -- ONE random 3-colouring instance of N x 100k variables (same degree,
same cost convention) is partitioned across the ranks, 100k variables per GPU, and
boundary V->F messages cross once per cycle -- as xGMI peer stores from the variable
kernel into hipIpc-mapped ghost buffers (one fused launch per cycle, no collective), or,
if a shard does not qualify or the in-kernel waits expire in the warm-up, as an RCCL
all-to-all issued by the engine itself; the cycle loop stays in the library either way
(pydcop_amd/sharded.py; MAXSUM_COLLECTIVE=p2p|rccl|torch forces one).  `value` is then N x iterations/s: the whole job's
throughput in iterations of a 100k-variable instance (= directed edge-messages/s
divided by the 800k messages of one such iteration), so N = 1 is the plain metric.
`--scaling strong` keeps the fixed 100k instance and splits it instead.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
# the launch(es) one cycle is made of, per workload (roofline.avg_launch_us covers them all)
KERNEL_OF = {"meeting_50k": "k_factor_nary + k_variable_wide (one cycle)"}
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")


def measured_traffic(workload, dtype):
    """HBM-side bytes per launch of the dominant kernel from the rocprofv3 PMC
    passes (FETCH_SIZE + WRITE_SIZE, separate runs; scripts/collect_traffic.py
    turns the committed profiles/*.csv into profiles/traffic.json).  None when
    this workload/dtype was not profiled."""
    try:
        with open(TRAFFIC_FILE) as f:
            t = json.load(f)
        return t.get(f"{workload}/{dtype}", {}).get("bytes_per_launch")
    except (OSError, ValueError):
        return None


def make_workload(name, n_gpus=1, per_gpu=100_000):
    from pydcop_amd import generators as G
    if name == "coloring_100k":     # the metric's configuration (north-star); x n_gpus when weak-scaled
        return G.random_coloring(per_gpu * n_gpus, avg_degree=4, n_colors=3, seed=0, names=False), "min"
    if name == "coloring_10k":      # BASELINE.json configs[1]
        return G.random_coloring(10_000, avg_degree=4, n_colors=3, seed=0, names=False), "min"
    if name == "coloring_100k_hard":
        return G.random_coloring(100_000, seed=0, variant="hard", names=False), "min"
    if name == "ising_1024":        # configs[2]
        return G.ising_grid(1024, 1024, seed=0, names=False), "min"
    if name == "coloring_1m_deg6":  # configs[3]
        return G.random_coloring(1_000_000, avg_degree=6, n_colors=3, seed=0, names=False), "min"
    if name == "meeting_50k":       # configs[4]
        return G.meeting_like(50_000, dom=24, arity=3, seed=0, names=False), "max"
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(graph, mode, dtype, budget_s=12.0):
    """The oracle (plain-C port of the reference algorithm) timed on the host
    cores on a bounded number of cycles of the same workload.  The thread count
    is the fastest of a few candidates (a 100k-variable cycle is too short to
    feed every core of a big host)."""
    from oracle.maxsum_oracle import OracleMaxSum, build
    from pydcop_amd.graph import Params
    build()
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    best = None
    for th in sorted({t for t in (1, 4, 8, 16, 32, 64, avail) if t <= avail}):
        ora = OracleMaxSum(graph, Params(mode=mode, dtype=dtype), threads=th)
        ora.run(1)  # warm
        t0 = time.perf_counter()
        ora.run(2)
        per = (time.perf_counter() - t0) / 2
        ora.close()
        if best is None or per < best[1]:
            best = (th, per)
    cores, per = best
    ora = OracleMaxSum(graph, Params(mode=mode, dtype=dtype), threads=cores)
    ora.run(1)
    n = int(max(3, min(2000, budget_s / max(per, 1e-6))))
    t0 = time.perf_counter()
    ora.run(n)
    dt = time.perf_counter() - t0
    ora.close()
    return {"value": n / dt, "unit": "iterations/s", "cores": cores, "kind": "port",
            "sample": f"{n} cycles of the same instance, oracle/maxsum_oracle.c (OpenMP, best of "
                      f"1..{avail} threads = {cores}; host has {os.cpu_count()} logical cpus)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="coloring_100k")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--layout-flags", type=int, default=0)
    ap.add_argument("--graph-chunk", type=int, default=-1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = N x 100k-variable instance (default), strong = the fixed instance")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="N > 1: torch.distributed backend; gloo only for the CPU test of this script "
                         "(tests/test_bench_cli.py, emulated engine)")
    ap.add_argument("--vars-per-gpu", type=int, default=100_000,
                    help="testing only: size of the coloring_100k workload (the metric is defined at 100000)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N > 1 with "
                         "python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")

    if world > 1:
        # dmabuf IPC (hipIpc handles of the peer-store exchange, RCCL's own buffers): has to be
        # in the environment before the ROCm runtime starts; normally exported already
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # one HIP runtime per process: torch first, so that the engine binds to the
        # runtime torch bundles and can share its stream / buffers with RCCL
        import torch
    from pydcop_amd.engine import MaxSumEngine
    from pydcop_amd.graph import Params

    weak = args.scaling == "weak" and args.gpus > 1 and args.workload == "coloring_100k"
    graph, mode = make_workload(args.workload, args.gpus if weak else 1, args.vars_per_gpu)
    units = args.gpus if weak else 1  # 100k-variable instances' worth of work per iteration
    params = Params(mode=mode, dtype=args.dtype, layout_flags=args.layout_flags,
                    graph_chunk=args.graph_chunk)
    word = 8 if args.dtype == "f64" else 4
    n_edges_total = graph.n_edges

    if world > 1:
        from pydcop_amd.sharded import ShardedMaxSum
        import torch.distributed as dist
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            # one node: RCCL bootstraps over loopback, no InfiniBand probing (the container's
            # hostname may not resolve); respected only if the launcher did not set them
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
            os.environ.setdefault("NCCL_IB_DISABLE", "1")
        dist.init_process_group(args.backend)
        dev = local_rank if args.backend == "nccl" else 0
        runner = ShardedMaxSum(graph, params, rank, world, device=dev)
        barrier = dist.barrier
        if runner.collective == "p2p":
            # the peer-store exchange has in-kernel waits with a time limit; if they expire on
            # this node (reported by the engine at sync), every rank falls back to RCCL together
            from pydcop_amd.engine import MaxSumGpuError
            ok = 1
            try:
                runner.run(min(args.warmup, 20) or 1)
            except MaxSumGpuError as e:
                ok = 0
                print(f"[rank {rank}] peer-store exchange failed ({e}); falling back to RCCL", file=sys.stderr)
            flag = torch.tensor([ok], device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                runner.close()
                runner = ShardedMaxSum(graph, params, rank, world, device=dev, collective="rccl")
    else:
        runner = MaxSumEngine(graph, params, device=local_rank)
        barrier = lambda: None  # noqa: E731

    def sync():
        runner.sync()  # hipStreamSynchronize on the engine's streams
        if world > 1 and args.backend == "nccl":
            torch.cuda.synchronize()

    runner.run(args.warmup)
    sync()
    barrier()
    t0 = time.perf_counter()
    if world > 1:
        runner.run(args.steps)
        event_ms = None
    else:
        event_ms = runner.run_timed(args.steps)  # HIP events on the engine's stream
    sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    check = None
    if world > 1:
        # outside the timed region: the sharded run must select what ONE engine sweeping the whole
        # instance selects after the same number of cycles (bit-identical beliefs too)
        idx_sh, bel_sh = runner.assignment()
        n_cycles = runner.cycle_count
        if rank == 0:
            import numpy as np
            with MaxSumEngine(graph, params, device=dev) as whole:
                whole.run(n_cycles)
                idx_1, bel_1 = whole.assignment()
            diff = int((idx_sh != idx_1).sum()) + int((bel_sh != bel_1).sum())
            check = {"cycles": int(n_cycles), "identical_to_single_engine": diff == 0, "differences": diff}
            if diff:
                print(f"[bench] sharded run differs from the single engine in {diff} places", file=sys.stderr)

    if rank == 0:
        its = units * args.steps / elapsed
        bytes_cycle = graph.cycle_bytes(word)
        out = {
            "metric": "MaxSum iterations/sec on 100k-var random graph-coloring DCOP",
            "value": its, "unit": "iterations/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak" if (weak or args.gpus == 1) else "strong",
            "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": args.workload + (f" x{args.gpus} (one {graph.n_vars}-variable instance, "
                                                    f"100k variables per GPU)" if weak else ""),
                       "n_vars": graph.n_vars,
                       "n_factors": graph.n_factors, "n_edges": graph.n_edges,
                       "domain": int(graph.dom_size.max()),
                       "edge_messages_per_s": args.steps / elapsed * 2 * n_edges_total,
                       "params": "damping 0.5/both, stability 0.1, start leafs",
                       "parallelism": f"graph-partition x{args.gpus}"
                                      + (f", exchange: {runner.collective}" if world > 1 else "")},
        }
        if event_ms is not None:
            kernel_s = event_ms * 1e-3 / args.steps
            achieved = bytes_cycle / kernel_s / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                               "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                               "traffic": measured_traffic(args.workload, args.dtype),
                               "kernel": KERNEL_OF.get(args.workload, "k_sweep"),
                               "algorithmic_bytes_per_launch": bytes_cycle,
                               "avg_launch_us": kernel_s * 1e6}
        else:  # N > 1: per-GPU figure from the wall clock of the whole sharded cycle
            per_gpu = bytes_cycle / args.gpus / (elapsed / args.steps) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": per_gpu, "peak": HBM_PEAK_GBPS,
                               "unit": "GB/s", "frac": per_gpu / HBM_PEAK_GBPS, "traffic": None,
                               "kernel": {"p2p": "k_sweep_p2p (one fused launch, peer stores over xGMI) + k_p2p_publish",
                                          "rccl": "k_sweep x2 + RCCL all-to-all issued by the engine (one sharded cycle)",
                                          "torch": "k_sweep x2 + halo pack/unpack + torch all_to_all_single"
                                          }.get(runner.collective, "one sharded cycle"),
                               "algorithmic_bytes_per_launch": bytes_cycle // args.gpus,
                               "avg_launch_us": 1e6 * elapsed / args.steps, "per_gpu": True}
        if check is not None:
            out["config"]["check"] = check
        if args.gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(graph, mode, args.dtype)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
