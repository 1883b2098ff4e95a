#!/bin/bash
# Round 5, mid-round validation: the whole GPU suite, the shard prediction with the cut classes beside launch 1
# (and behind it: MAXSUM_SHARD_CUT_BESIDE=0), the default bench line.
TAG=${1:-r5_mid}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== shard prediction (N = 8), beside / behind"
for b in 1 0; do
  MAXSUM_SHARD_CUT_BESIDE=$b timeout 600 python tools/scale_prediction.py --ranks 8 > $OUT/pred_beside$b.log 2>&1
  grep '^{"n"' $OUT/pred_beside$b.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('beside=$b', 'compute', round(d['shard_compute_us'],1), 'loopback', round(d.get('shard_cycle_us_rccl_loopback',-1),1), d.get('predicted_speedup_vs_one_gpu'))"
done
echo "== sharded GPU tests"
( time timeout 900 python -m pytest tests/test_sharded.py -x -q -m gpu ) 2>&1 | tail -4 | tee $OUT/pytest_sharded.txt
echo "== GPU suite"
( time timeout 2400 python -m pytest tests -x -q -m gpu ) 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
echo "== default bench line"
timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 300 $OUT/bench_default.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print("value", round(d["value"],1), "ms/step", d["ms_per_step"], "timing", d["timing"])
print("roofline", {k: d["roofline"][k] for k in ("frac","avg_launch_us","frac_of_stored_bytes","frac_by_traffic") if k in d["roofline"]})
for c in d.get("configs", []):
    r=c["roofline"]; print("  %-20s %s %8.1f us frac %.3f stored %.3f %s" % (c["workload"], c["dtype"], c["ms_per_step"]*1e3, r["frac"], r.get("frac_of_stored_bytes",0), c.get("factor_kernels")))
PY
exit 0
