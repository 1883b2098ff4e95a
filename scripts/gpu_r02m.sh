#!/bin/bash
# round 2, call M: DSA / MGM on the GPU
TAG=${1:-r02m}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_dsa.py tests/test_gpu_mgm.py -x -q -m gpu --durations=4 2>&1 | tail -12 | tee $OUT/pytest.txt
python - <<'PY' | tee $OUT/local_search_bench.jsonl
import json, time
from pydcop_amd import generators as G
from pydcop_amd.dsa import DsaEngine
from pydcop_amd.mgm import MgmEngine
from pydcop_amd.graph import Params
g = G.random_coloring(100_000, seed=0, names=False)
for name, eng in (("dsa_B", DsaEngine(g, Params(), variant="B", seed=1)), ("mgm", MgmEngine(g, Params()))):
    eng.run(20)
    t0 = time.perf_counter(); eng.run(500); dt = time.perf_counter() - t0
    print(json.dumps({"algo": name, "n_vars": g.n_vars, "cycles_per_s": round(500 / dt, 1), "us_per_cycle": round(1e6 * dt / 500, 2),
                      "cost": eng.eval_cost()[0]}))
    eng.close()
PY
