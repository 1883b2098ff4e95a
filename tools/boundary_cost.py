"""Host-boundary cost of a one-shot solve (not the bench metric): flat arrays in host
memory -> mxs_create (layout build + PCIe upload + cycle 0) -> T cycles -> assignment back.
usage: python tools/boundary_cost.py [workload] [T]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_workload
from pydcop_amd.engine import MaxSumEngine
from pydcop_amd.graph import Params

w = sys.argv[1] if len(sys.argv) > 1 else "coloring_100k"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
g, mode = make_workload(w)
MaxSumEngine(g, Params(mode=mode)).close()          # warm the runtime / code objects
t0 = time.perf_counter(); e = MaxSumEngine(g, Params(mode=mode)); t1 = time.perf_counter()
e.run(T); t2 = time.perf_counter()
idx, belief = e.assignment(); t3 = time.perf_counter()
host_bytes = sum(a.nbytes for a in (g.dom_size, g.var_cost, g.factor_rowptr, g.edge_var, g.table_off,
                                    g.tables, g.var_rowptr, g.var_edges))
print(json.dumps({"workload": w, "cycles": T, "host_graph_bytes": host_bytes,
                  "create_ms": 1e3 * (t1 - t0), "run_ms": 1e3 * (t2 - t1), "download_ms": 1e3 * (t3 - t2),
                  "iterations_per_s_incl_boundary": T / (t3 - t0),
                  "iterations_per_s_resident": T / (t2 - t1)}))
