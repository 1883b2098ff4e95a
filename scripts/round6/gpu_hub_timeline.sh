#!/bin/bash
# Round 6: per-block timeline of the sweep launch on the scale-free instance: which class holds the cycle?
TAG=${1:-r6_hub_tl}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for wl in coloring_100k_scalefree coloring_1m_scalefree; do
  timeout 300 python3 tools/timeline.py $wl f64 2>&1 | cut -c1-400 | tee $OUT/timeline_$wl.txt | grep -v "active blocks" | tail -12
done
timeout 300 python3 - <<'PY' 2>&1 | tee $OUT/longest_blocks.txt
import sys, numpy as np
sys.path.insert(0, '.')
from bench import make_workload
from pydcop_amd.engine import MaxSumEngine
from pydcop_amd.graph import Params
g, mode = make_workload("coloring_100k_scalefree")
deg = np.diff(g.var_rowptr)
print("degree histogram: >32", (deg > 32).sum(), ">64", (deg > 64).sum(), "33..64", ((deg > 32) & (deg <= 64)).sum(), "17..32", ((deg > 16) & (deg <= 32)).sum())
e = MaxSumEngine(g, Params(mode=mode, graph_chunk=0))
e.run(50)
t = e.debug_timeline()
t0 = t[:, 0].min()
s, f, k = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, t[:, 2]
order = np.argsort(-(f - s))[:25]
for b in order:
    print("block", int(b), "kind", int(k[b]), "start %.2f end %.2f dur %.2f" % (s[b], f[b], f[b] - s[b]))
PY
exit 0
