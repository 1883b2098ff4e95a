import sys, time
sys.path.insert(0, '/root/repo')
from pydcop_amd import generators as G
from pydcop_amd.amaxsum import AMaxSumEngine
from pydcop_amd.graph import Params
g = G.random_coloring(100000, avg_degree=4, n_colors=3, seed=0, names=False)
t0=time.perf_counter(); eng = AMaxSumEngine(g, Params(start_messages="leafs_vars")); print("create", round(time.perf_counter()-t0,4))
for gen in range(1, 17):
    t0=time.perf_counter(); d=eng.run(gen); print(gen, d, round((time.perf_counter()-t0)*1e3,2), "ms")
