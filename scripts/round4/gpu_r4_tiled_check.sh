#!/bin/bash
# tiled factor order as the default rule: parity tests that exercise it, the metric's line, the other configs
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_tiled_check; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "tiled or layout_variants or north_star or config2" 2>&1 | tail -4 | tee $OUT/tests.txt
timeout 600 python -m pytest tests/test_sharded.py tests/test_gpu_fuzz.py -q -m gpu 2>&1 | tail -3 | tee -a $OUT/tests.txt
timeout 600 python bench.py --no-cpu-baseline --configs all 2>&1 | grep '^{' > $OUT/bench.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_tiled_check/bench.json").read())
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"].get("factor_order"))
for c in d.get("configs", []):
    print(c["workload"], c["dtype"], round(c["ms_per_step"]*1000,2), round(c["roofline"]["frac"],4), c.get("factor_order"))
PY
