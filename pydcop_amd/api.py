"""Direct API: solve a pyDCOP `DCOP` object (or a YAML file, or an .npz instance) on the GPU
without agents -- synchronous Max-Sum by default, `algo=` "amaxsum", "dsa" or "mgm" for the other
engines of the library.

`pydcop.infrastructure.run.solve` (pydcop/infrastructure/run.py:49) deploys one
computation per node on agent threads and lets an orchestrator collect the values;
with `maxsum_gpu` those computations are proxies and the work is one engine anyway
(pydcop_amd/algorithms/maxsum_gpu.py).  This entry point goes straight from the DCOP
to the engine -- O(E) graph build, flat arrays, T cycles, values back -- and reports
with the reference's own `DCOP.solution_cost` (pydcop/dcop/dcop.py:308-367).
pyDCOP must be importable; nothing of it is modified.
"""
from typing import Dict, List, Optional, Tuple

import numpy as np

from .compile import assignment_to_values, compile_nodes
from .graph import FlatGraph, Params


ALGOS = ("maxsum", "amaxsum", "dsa", "mgm")


def _engine_for(graph: FlatGraph, params: Params, device: int, devices: int, lib_path, algo: str = "maxsum",
                algo_kw: Optional[dict] = None):
    """One engine on `device`, or -- Max-Sum with devices > 1 -- the graph partitioned over GPUs
    0..devices-1 of this node (pydcop_amd.sharded.LocalShardedMaxSum: same surface, same result).
    Every engine has run / assignment / eval_cost / close."""
    algo_kw = algo_kw or {}
    if algo not in ALGOS:
        raise ValueError(f"algo must be one of {ALGOS}")
    if algo != "maxsum":
        if devices and int(devices) > 1:
            raise ValueError("devices > 1: synchronous Max-Sum only")
        if algo == "amaxsum":
            from .amaxsum import AMaxSumEngine
            return AMaxSumEngine(graph, params, device=device, lib_path=lib_path)
        if algo == "dsa":
            from .dsa import DsaEngine
            return DsaEngine(graph, params, device=device, lib_path=lib_path, **algo_kw)
        from .mgm import MgmEngine
        return MgmEngine(graph, params, device=device, lib_path=lib_path)
    from .engine import MaxSumEngine
    if devices and int(devices) > 1:
        from .sharded import LocalShardedMaxSum
        return LocalShardedMaxSum(graph, params, list(range(int(devices))), lib_path=lib_path)
    return MaxSumEngine(graph, params, device=device, lib_path=lib_path)


def _run_and_trace(eng, algo: str, cycles: int, cost_every: int, infinity: float):
    """`cycles` cycles (amaxsum: generations of messages, its run() counts from the start), the
    cost evaluated on the device every `cost_every` of them."""
    curve: List[Tuple[int, float, int]] = []
    done = 0
    while done < cycles:
        n = min(cost_every, cycles - done) if cost_every > 0 else cycles - done
        eng.run(done + n if algo == "amaxsum" else n)
        done += n
        if cost_every > 0:
            c, v = eng.eval_cost(infinity=infinity)
            curve.append((done, c, v))
    return curve


def compile_dcop(dcop, noise: float = 0.0, seed: int = 0) -> FlatGraph:
    """DCOP -> FlatGraph in the node / links order the reference's factor graph has
    (pydcop/computations_graph/factor_graph.py:245-296), built in O(E)."""
    from . import plugin
    plugin.install()
    from pydcop.computations_graph import factor_graph_fast
    cg = factor_graph_fast.build_computation_graph(dcop)
    var_nodes = [n for n in cg.nodes if n.type == "VariableComputation"]
    factor_nodes = [n for n in cg.nodes if n.type == "FactorComputation"]
    return compile_nodes(var_nodes, factor_nodes, noise=noise, rng=np.random.default_rng(seed))


def solve_dcop(dcop, cycles: int = 30, *, damping: float = 0.5, damping_nodes: str = "both",
               stability: float = 0.1, noise: float = 0.01, start_messages: str = "leafs",
               precision: str = "f64", seed: int = 0, infinity: float = 10000, device: int = 0,
               cost_every: int = 0, lib_path: Optional[str] = None, devices: int = 1, algo: str = "maxsum",
               variant: str = "B", probability: float = 0.7, p_mode: str = "fixed") -> Dict:
    """Synchronous Max-Sum for exactly `cycles` cycles; parameters and defaults are those
    of `pydcop.algorithms.maxsum` (maxsum.py:212-220), `infinity` that of
    `pydcop.infrastructure.run.solve` (run.py:49).  `algo`: "amaxsum" (`cycles` = generations of
    messages under FIFO delivery, same parameters), "dsa" (`variant`, `probability`, `p_mode` of
    pydcop.algorithms.dsa, dsa.py:119-125; `seed` keys its draws) or "mgm"; the local-search
    algorithms take no noise (their variable costs enter as the reference's do).

    Returns {"assignment", "cost", "violation", "cycle", "cost_curve"}: the first three
    as `DCOP.solution_cost` computes them for the selected values; `cost_curve` (when
    `cost_every` > 0) = [(cycle, cost, violations)] evaluated on the device every
    `cost_every` cycles (the reference's `--collect_on cycle_change`,
    pydcop/commands/solve.py:356-376, without leaving the GPU)."""
    graph = compile_dcop(dcop, noise=noise if algo in ("maxsum", "amaxsum") else 0.0, seed=seed)
    params = Params(mode=dcop.objective, damping=damping, damping_nodes=damping_nodes,
                    stability=stability, start_messages=start_messages, dtype=precision)
    algo_kw = dict(variant=variant, probability=probability, p_mode=p_mode, seed=seed) if algo == "dsa" else None
    with _engine_for(graph, params, device, devices, lib_path, algo, algo_kw) as eng:
        curve = _run_and_trace(eng, algo, cycles, cost_every, infinity)
        idx, _ = eng.assignment()
    assignment = assignment_to_values(graph, idx)
    violation, cost = dcop.solution_cost(assignment, infinity)
    return {"assignment": assignment, "cost": cost, "violation": violation, "cycle": cycles,
            "cost_curve": curve}


def solve_flat(graph: FlatGraph, objective: str = "min", cycles: int = 30, *, damping: float = 0.5,
               damping_nodes: str = "both", stability: float = 0.1, start_messages: str = "leafs",
               precision: str = "f64", infinity: float = 10000, device: int = 0, cost_every: int = 0,
               lib_path: Optional[str] = None, devices: int = 1, algo: str = "maxsum", variant: str = "B",
               probability: float = 0.7, p_mode: str = "fixed", seed: int = 0) -> Dict:
    """`solve_dcop` for an already compiled instance (`FlatGraph`, e.g. loaded from the
    .npz instance format): no pyDCOP import at all.  Cost and violations come from the
    device (`mxs_eval_cost` = DCOP.solution_cost, pydcop/dcop/dcop.py:308-367); noise, if
    wanted, is already folded into `graph.var_cost` by whoever compiled the instance."""
    params = Params(mode=objective, damping=damping, damping_nodes=damping_nodes,
                    stability=stability, start_messages=start_messages, dtype=precision)
    algo_kw = dict(variant=variant, probability=probability, p_mode=p_mode, seed=seed) if algo == "dsa" else None
    with _engine_for(graph, params, device, devices, lib_path, algo, algo_kw) as eng:
        curve = _run_and_trace(eng, algo, cycles, cost_every, infinity)
        idx, _ = eng.assignment()
        cost, violation = eng.eval_cost(infinity=infinity)
    if graph.var_names is not None and graph.domains is not None:
        assignment = assignment_to_values(graph, idx)
    else:
        assignment = {f"v{i}": int(x) for i, x in enumerate(idx)}
    return {"assignment": assignment, "cost": cost, "violation": violation, "cycle": cycles,
            "cost_curve": curve}


def solve_yaml(paths, cycles: int = 30, **kw) -> Dict:
    """`solve_dcop` on DCOP YAML file(s) (pydcop/dcop/yamldcop.py:96)."""
    from . import plugin
    plugin.install()
    from pydcop.dcop.yamldcop import load_dcop_from_file
    if isinstance(paths, str):
        paths = [paths]
    return solve_dcop(load_dcop_from_file(list(paths)), cycles, **kw)


def main(argv=None):
    """`python -m pydcop_amd.api [-c CYCLES] [-p name:value ...] dcop.yaml ... | instance.npz`
    -- solve without agents and print a result in the schema of `pydcop solve`
    (docs/tutorials/analysing_results.rst:31-48; no agent metrics: there are no agents).
    `--export out.npz` compiles the YAML DCOP to the binary instance format instead
    (`FlatGraph.save`); an .npz instance is solved without importing pyDCOP."""
    import argparse
    import json
    import os
    import time
    ap = argparse.ArgumentParser(prog="python -m pydcop_amd.api")
    ap.add_argument("dcop_files", nargs="+")
    ap.add_argument("-c", "--cycles", type=int, default=30)
    ap.add_argument("-a", "--algo", default="maxsum", choices=ALGOS)
    ap.add_argument("-p", "--algo_params", action="append", default=[],
                    help="name:value, e.g. damping:0.7 noise:0 precision:f32 (maxsum.py:212-220)")
    ap.add_argument("--infinity", type=float, default=float("inf"))   # pydcop/commands/solve.py:316-324
    ap.add_argument("--cost_every", type=int, default=0)
    ap.add_argument("--export", metavar="OUT.npz", default=None,
                    help="compile the YAML DCOP (noise folded in) and write it as an instance file")
    args = ap.parse_args(argv)
    kinds = {"damping": float, "stability": float, "noise": float, "seed": int,
             "damping_nodes": str, "start_messages": str, "precision": str, "devices": int,
             "variant": str, "probability": float, "p_mode": str}
    kw = {}
    for item in args.algo_params:
        name, _, value = item.partition(":")
        if name not in kinds:
            raise SystemExit(f"Error: unknown parameter {name!r} (one of {sorted(kinds)})")
        kw[name] = kinds[name](value)
    t0 = time.perf_counter()
    if args.export:
        from . import plugin
        plugin.install()
        from pydcop.dcop.yamldcop import load_dcop_from_file
        dcop = load_dcop_from_file(list(args.dcop_files))
        g = compile_dcop(dcop, noise=kw.get("noise", 0.01), seed=kw.get("seed", 0))
        g.save(args.export, objective=dcop.objective, source=[os.path.basename(f) for f in args.dcop_files])
        print(json.dumps({"exported": args.export, "n_vars": g.n_vars, "n_factors": g.n_factors,
                          "n_edges": g.n_edges, "objective": dcop.objective}))
        return
    if len(args.dcop_files) == 1 and args.dcop_files[0].endswith(".npz"):
        graph, header = FlatGraph.load(args.dcop_files[0])
        kw.pop("noise", None)  # folded into the instance when it was compiled
        if args.algo != "dsa":
            kw.pop("seed", None)
        res = solve_flat(graph, header.get("objective", "min"), args.cycles, infinity=args.infinity,
                         cost_every=args.cost_every, algo=args.algo, **kw)
    else:
        res = solve_yaml(args.dcop_files, args.cycles, infinity=args.infinity,
                         cost_every=args.cost_every, algo=args.algo, **kw)
    out = {"assignment": res["assignment"], "cost": res["cost"], "violation": res["violation"],
           "cycle": res["cycle"], "status": "FINISHED", "time": time.perf_counter() - t0,
           "msg_count": 0, "msg_size": 0, "agt_metrics": {}}
    if res["cost_curve"]:
        out["cost_curve"] = res["cost_curve"]
    print(json.dumps(out, sort_keys=True, indent="  "))


if __name__ == "__main__":
    main()
