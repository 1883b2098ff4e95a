"""Per-block timeline of one sweep launch (profiling): where does a cycle's time go?
usage: python tools/timeline.py [workload] [dtype]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import make_workload
from pydcop_amd.engine import MaxSumEngine
from pydcop_amd.graph import Params

w = sys.argv[1] if len(sys.argv) > 1 else "coloring_100k"
dt = sys.argv[2] if len(sys.argv) > 2 else "f64"
g, mode = make_workload(w)
e = MaxSumEngine(g, Params(mode=mode, dtype=dt, graph_chunk=0))
e.run(50)
for rep in range(3):
    t = e.debug_timeline()
    t0 = t[:, 0].min()
    s, f, k = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, t[:, 2]   # microseconds
    print(f"rep {rep}: blocks {len(t)}  span {f.max():.2f} us")
    for kind in np.unique(k):
        m = k == kind
        print(f"  kind {kind}: n {m.sum():5d} start [{s[m].min():6.2f} .. {np.percentile(s[m],50):6.2f} .. {s[m].max():6.2f}] "
              f"end [{f[m].min():6.2f} .. {np.percentile(f[m],50):6.2f} .. {f[m].max():6.2f}] "
              f"dur p10 {np.percentile((f-s)[m],10):5.2f} p50 {np.percentile((f-s)[m],50):5.2f} p90 {np.percentile((f-s)[m],90):5.2f} max {(f-s)[m].max():5.2f}")
    # how many blocks are running over time
    grid = np.arange(0, f.max(), 1.0)
    act = [(int(((s <= x) & (f > x)).sum())) for x in grid]
    print("  active blocks per us:", act[:60])
