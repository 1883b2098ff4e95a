#!/bin/bash
# Round 6: a hub alone on the GPU (star graphs): duration of its workgroups by degree -- fixed latency vs per-element cost.
TAG=${1:-r6_hub_alone}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 300 python3 - <<'PY' 2>&1 | tee $OUT/hub_alone.txt
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from pydcop_amd.engine import MaxSumEngine
from pydcop_amd.graph import Params, FlatGraph
def star(deg, D=3, seed=0):
    rng = np.random.default_rng(seed)
    n = deg + 1
    edge_var = np.stack([np.zeros(deg, int), np.arange(1, n)], 1).reshape(-1).astype(np.int32)
    rowptr = (2 * np.arange(deg + 1)).astype(np.int32)
    tables = rng.integers(0, 10, deg * D * D).astype(float)
    toff = (D * D * np.arange(deg + 1)).astype(np.int64)
    dom = np.full(n, D, dtype=np.int32)
    cost = rng.uniform(0, 0.01, n * D)
    vr, ve = FlatGraph.var_side_from_edges(edge_var, n)
    return FlatGraph(dom_size=dom, var_cost=cost, factor_rowptr=rowptr, edge_var=edge_var, table_off=toff, tables=tables, var_rowptr=vr, var_edges=ve).validate()
for dt in ("f64", "f32"):
    for deg in (70, 200, 705, 1400, 2100, 4200):
        g = star(deg)
        e = MaxSumEngine(g, Params(dtype=dt, graph_chunk=0))
        e.run(20)
        best = None
        for rep in range(5):
            t = e.debug_timeline()
            t0 = t[:, 0].min()
            s, f, k = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, t[:, 2]
            h = k == 9
            d = (f - s)[h].max()
            best = d if best is None else min(best, d)
        print(dt, "deg", deg, "hub blocks", int(h.sum()), "longest hub block %.2f us" % best, "elements", 3 * deg, "ns/element %.2f" % (1e3 * best / (3 * deg)))
PY
exit 0
