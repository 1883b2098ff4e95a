"""Front-end timing (CPU, needs the reference checkout): DCOP objects -> computation graph
-> flat arrays, reference builder vs factor_graph_fast.
usage: python tools/frontend_bench.py N [--ref]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.environ.get("PYDCOP_REFERENCE", "/root/reference"))
import numpy as np
from pydcop_amd import plugin
plugin.install()
from pydcop.computations_graph import factor_graph, factor_graph_fast
from pydcop.dcop.dcop import DCOP
from pydcop.dcop.objects import Domain, VariableWithCostDict
from pydcop.dcop.relations import NAryMatrixRelation
from pydcop_amd.compile import compile_computation_graph

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(0)
d = Domain("colors", "", [0, 1, 2])
t0 = time.perf_counter()
dcop = DCOP("bench", objective="min")
vs = [VariableWithCostDict(f"v{i:06d}", d, {k: float(rng.uniform(0, 0.01)) for k in range(3)}) for i in range(n)]
for v in vs:
    dcop.variables[v.name] = v
dcop.domains[d.name] = d
pairs = set()
while len(pairs) < 2 * n:
    a, b = rng.integers(0, n, 2)
    if a != b:
        pairs.add((min(a, b), max(a, b)))
for k, (a, b) in enumerate(sorted(pairs)):
    c = NAryMatrixRelation([vs[a], vs[b]], rng.integers(0, 10, (3, 3)).astype(float), name=f"c{k:06d}")
    dcop.constraints[c.name] = c
t1 = time.perf_counter()
out = {"n_vars": n, "n_constraints": len(pairs), "build_dcop_objects_s": t1 - t0}
if "--ref" in sys.argv:
    t = time.perf_counter(); factor_graph.build_computation_graph(dcop); out["reference_build_computation_graph_s"] = time.perf_counter() - t
t = time.perf_counter(); cg = factor_graph_fast.build_computation_graph(dcop); out["factor_graph_fast_s"] = time.perf_counter() - t
t = time.perf_counter(); g = compile_computation_graph(cg); out["compile_to_flat_arrays_s"] = time.perf_counter() - t
out["n_edges"] = g.n_edges
print(json.dumps(out))
