"""The reference's own Max-Sum computations (pydcop/algorithms/maxsum.py) timed on the host
CPU for a fixed number of cycles: instances of the benchmark family built by our O(E)
generator, converted to pyDCOP objects, driven by the single-thread FIFO harness of
oracle/ref_harness.py (no agents, queues or orchestrator: an UPPER bound of what the
reference's thread-agent path reaches).  Needs the reference checkout
(build container only).  usage: python tools/reference_cpu_baseline.py [n_vars ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as R
from pydcop_amd import generators as G

if not R.reference_available():
    raise SystemExit("the pyDCOP reference checkout is not on this machine")
R.install_shims()
for n in [int(x) for x in sys.argv[1:]] or [1000, 10000]:
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
    print(json.dumps({"n_vars": n, "n_factors": g.n_factors, "n_edges": g.n_edges, "cycles": cycles,
                      "to_pydcop_objects_s": round(t1 - t0, 2), "setup_plus_one_cycle_s": round(t2 - t1, 2),
                      "seconds_per_cycle": round(per_cycle, 3), "iterations_per_s": round(1 / per_cycle, 4),
                      "edge_messages_per_s": round(2 * g.n_edges / per_cycle, 1),
                      "cpu": "1 thread (the reference is pure Python, GIL-bound)"}), flush=True)
