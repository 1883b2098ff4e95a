#!/bin/bash
# Round 5 closing call: the full GPU suite, smoke, the default bench line, the DRIVER's command (--steps 20 --warmup 5),
# kernel traces of the metric's configuration under both commands.
TAG=${1:-r5_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== pytest -m gpu (everything)"
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 ) 2>&1 | tail -24 | tee $OUT/pytest_gpu.txt
echo "== smoke"
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -6 | tee $OUT/smoke.txt
echo "== bench default"
( time timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | tail -3
echo "== the driver's command: --gpus 1 --steps 20 --warmup 5"
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 --configs main --reference-budget 0 > $OUT/bench_driver.json 2> $OUT/bench_driver.err ) 2>&1 | tail -3
python - <<PY
import json
for n in ("bench_default", "bench_driver"):
    d=json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
    print(n, "value", round(d["value"],1), "ms/step", round(d["ms_per_step"]*1e3,3), "us; repeats", d["timing"]["repeats"], "min/max", round(d["timing"]["ms_per_step_min"]*1e3,3), round(d["timing"]["ms_per_step_max"]*1e3,3), "wall", round(d["timing"]["wall_ms_per_step_over_the_region"]*1e3,3), "frac", round(d["roofline"]["frac"],4))
    for c in d.get("configs", []):
        r=c["roofline"]; print("  %-20s %s %8.1f us frac %.3f stored %.3f traffic %s %s %s" % (c["workload"], c["dtype"], c["ms_per_step"]*1e3, r["frac"], r.get("frac_of_stored_bytes",0), r.get("frac_by_traffic"), c.get("factor_kernels"), ""))
    for a in d.get("algorithms", []):
        print("  ", a["algo"][:40], {k: a[k] for k in a if k in ("us_per_cycle","messages_per_s","cycles_per_s")})
PY
echo "== rocprofv3 kernel trace of the metric's configuration (default command, then the driver's)"
cd /tmp
for spec in "default:--steps 2000 --warmup 200" "driver:--steps 20 --warmup 5"; do
  name=${spec%%:*}; args=${spec#*:}
  rm -rf $OUT/p
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python $R/bench.py --no-cpu-baseline --configs main $args > $OUT/prof_$name.log 2>&1
  f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_coloring100k_$name.csv && cut -c1-200 $OUT/kernel_stats_coloring100k_$name.csv | head -4
  rm -rf $OUT/p
done
echo "== the stated multi-GPU prediction, re-measured on this code"
cd $R
timeout 900 python tools/scale_prediction.py --out $OUT/scale_prediction.json > $OUT/scale_prediction.log 2>&1
grep '^{"n"' $OUT/scale_prediction.log | python -c "
import sys, json
for line in sys.stdin:
    d = json.loads(line); print('N', d['n'], 'compute', round(d['shard_compute_us'],1), 'loopback', round(d.get('shard_cycle_us_rccl_loopback',-1),1), d.get('predicted_speedup_vs_one_gpu'))"
exit 0
