"""The reference's own Max-Sum timed on the host CPU of whatever box this runs on (build
container: /root/reference; GPU box: the git-ignored archive oracle/_ref/ that build() packs,
see oracle/stage_reference.py).  bench.py's cpu_baseline leg calls `--mode threads` on a
1 000-variable instance; 10 000 variables cost minutes of wall time per run (the orchestrator
deploys 30 000 computations one message at a time) and are run on request only.

Two ways, both on instances of the benchmark family built by our O(E) generator and converted
to pyDCOP objects (oracle/ref_harness.flat_to_dcop):

  --mode threads   (the OFFICIAL path)  the reference's thread-agent
                   runtime: pydcop.infrastructure.run.run_local_thread_dcop (run.py:145) with k
                   agents (k in {1, nproc}), computations dealt to the agents in contiguous
                   blocks, orchestrator.deploy_computations() + run(timeout=T) exactly as
                   run.solve does (run.py:124-135); throughput = end_metrics()['cycle'] /
                   end_metrics()['time'] (orchestrator.py:1258-1270).
  --mode fifo      (8d(ii), an UPPER bound of the above)  the MaxSum computations driven by the
                   single-thread FIFO harness of oracle/ref_harness.py: no agents, queues or
                   orchestrator.

usage: python tools/reference_cpu_baseline.py [--mode threads|fifo] [--timeout T] [--agents k ...]
                                               [--out profiles/x.jsonl] [--n-vars n ...]
"""
import argparse
import json
import os
import platform
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as R  # noqa: E402
from pydcop_amd import generators as G  # noqa: E402


def host_info():
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"host": f"{platform.node()}: {os.cpu_count()} logical cpus, {model}; python {platform.python_version()}",
            "cores": len(os.sched_getaffinity(0))}


def run_fifo(n):
    g = G.random_coloring(n, avg_degree=4, n_colors=3, seed=0)
    t0 = time.perf_counter()
    dcop, cg = R.flat_to_dcop(g, "min")  # pyDCOP objects + factor graph, O(E)
    t1 = time.perf_counter()
    cycles = 5 if n <= 2000 else 2
    R.run_reference_maxsum(dcop, 1, cg=cg)          # builds the computations, one cycle
    t2 = time.perf_counter()
    R.run_reference_maxsum(dcop, 1 + cycles, cg=cg)
    t3 = time.perf_counter()
    per_cycle = ((t3 - t2) - (t2 - t1)) / cycles   # the second run repeats the set-up + first cycle
    return {"mode": "fifo", "n_vars": n, "n_factors": g.n_factors, "n_edges": g.n_edges, "cycles": cycles,
            "to_pydcop_objects_s": round(t1 - t0, 2), "setup_plus_one_cycle_s": round(t2 - t1, 2),
            "seconds_per_cycle": round(per_cycle, 3), "iterations_per_s": round(1 / per_cycle, 4),
            "edge_messages_per_s": round(2 * g.n_edges / per_cycle, 1),
            "cpu": "1 thread (the reference is pure Python, GIL-bound)"}


def run_threads(n, k, timeout):
    """run_local_thread_dcop with k agents; cycles / time as the orchestrator reports them."""
    import logging
    logging.disable(logging.CRITICAL)
    from pydcop.algorithms import AlgorithmDef
    from pydcop.dcop.objects import AgentDef
    from pydcop.distribution.objects import Distribution
    from pydcop.infrastructure.run import INFINITY, run_local_thread_dcop
    g = G.random_coloring(n, avg_degree=4, n_colors=3, seed=0)
    dcop, cg = R.flat_to_dcop(g, "min")
    names = [node.name for node in cg.nodes]
    agents = [AgentDef(f"a{i:03d}") for i in range(k)]
    dcop.add_agents(agents)
    per = (len(names) + k - 1) // k
    mapping = {a.name: names[i * per:(i + 1) * per] for i, a in enumerate(agents)}
    algo = AlgorithmDef.build_with_default_param("maxsum", {}, mode="min")
    # the agents are created inside run_local_thread_dcop and not handed back: remember them, so
    # that the cycle every computation reached can be read directly when the run ends (the
    # orchestrator's own figure is max over computations of what the agents REPORTED -- an
    # isolated variable spins through cycles on its own, and a busy agent may not report at all)
    import pydcop.infrastructure.run as run_mod
    made = []

    class RecordingAgent(run_mod.OrchestratedAgent):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            made.append(self)
    orig_agent = run_mod.OrchestratedAgent
    run_mod.OrchestratedAgent = RecordingAgent
    t0 = time.perf_counter()
    try:
        orchestrator = run_local_thread_dcop(algo, cg, Distribution(mapping), dcop, INFINITY)
    finally:
        run_mod.OrchestratedAgent = orig_agent
    connected = {n.name for n in cg.nodes if n.neighbors}
    reached = []
    try:
        orchestrator.deploy_computations()
        t1 = time.perf_counter()
        orchestrator.run(timeout=timeout)
        t2 = time.perf_counter()
        for a in made:
            for c in a.computations():
                if c.name in connected:
                    reached.append(int(c.cycle_count))
        m = orchestrator.end_metrics()
    finally:
        try:
            orchestrator.stop_agents(5)
            orchestrator.stop()
        except Exception:
            pass
        logging.disable(logging.NOTSET)
    secs = float(m["time"]) or (t2 - t1)
    reached.sort()
    cycles = reached[len(reached) // 2] if reached else 0   # median over the connected computations
    return {"mode": "threads", "n_vars": n, "n_factors": g.n_factors, "n_edges": g.n_edges, "agents": k,
            "timeout_s": timeout, "status": m["status"], "time_s": round(secs, 3),
            "cycle_median": cycles, "cycle_min": reached[0] if reached else 0, "cycle_max": reached[-1] if reached else 0,
            "cycle_orchestrator": int(m["cycle"]),
            "deploy_s": round(t1 - t0, 2), "run_wall_s": round(t2 - t1, 2),
            "iterations_per_s": round(cycles / secs, 4) if secs > 0 else None,
            "edge_messages_per_s": round(cycles * 2 * g.n_edges / secs, 1) if secs > 0 else None,
            "msg_count": m["msg_count"],
            "cpu": f"{k} agent thread(s) + orchestrator in one CPython process (GIL: about one core busy)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="threads", choices=["threads", "fifo"])
    ap.add_argument("--timeout", type=float, default=30.0)
    ap.add_argument("--agents", type=int, nargs="*", default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--n-vars", type=int, nargs="*", default=None,
                    help="instance sizes (default 1000; 10000 takes 6-20 minutes of wall time per run)")
    args = ap.parse_args()
    if not R.reference_available():
        raise SystemExit("the pyDCOP reference checkout is not on this machine")
    R.install_shims()
    info = host_info()
    sizes = args.n_vars or [1000]
    out = open(args.out, "a") if args.out else None
    for n in sizes:
        if args.mode == "fifo":
            recs = [run_fifo(n)]
        else:
            recs = [run_threads(n, k, args.timeout) for k in (args.agents or [1, info["cores"]])]
        for rec in recs:
            rec.update(info)
            line = json.dumps(rec)
            print(line, flush=True)
            if out:
                out.write(line + "\n")
                out.flush()


if __name__ == "__main__":
    main()
