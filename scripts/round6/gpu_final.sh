#!/bin/bash
# Round 6 closing call: the full GPU suite, smoke, the DRIVER's exact command with its raw stdout kept (the final line has to parse
# from the 8-KB tail), kernel traces of the metric's configuration under that command and of the new rows.
TAG=${1:-r6_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== pytest -m gpu (everything)"
( time timeout 2000 python3 -m pytest tests -x -q -m gpu --durations=8 ) 2>&1 | tail -24 | tee $OUT/pytest_gpu.txt
echo "== smoke"
timeout 300 python3 -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -6 | tee $OUT/smoke.txt
echo "== the driver's command, verbatim"
( time timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_stdout.txt 2> $OUT/bench_driver_stderr.txt ) 2>&1 | tail -3
tail -c 8192 $OUT/bench_driver_stdout.txt > $OUT/bench_driver_tail8k.txt
cp bench_rows.json $OUT/bench_rows.json 2>/dev/null
python3 - <<PY
import json
lines=open("$OUT/bench_driver_stdout.txt").read().strip().splitlines()
print("lines", len(lines), "last line bytes", len(lines[-1]))
d=json.loads(lines[-1]); r=d["roofline"]
print({k: d[k] for k in ("value","ms_per_step","n_gpus","steps","warmup")}, "frac", r["frac"], "stored", r.get("frac_of_stored_bytes"), "traffic", r.get("frac_by_traffic"), "cpu", d["cpu_baseline"]["value"])
for k, v in d["rows"].items(): print("  %-28s %9s us  frac %s" % (k, v[0], v[1]))
PY
echo "== rocprofv3 kernel trace of the metric's configuration under the driver's steps"
cd /tmp
rm -rf $OUT/p
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python3 $R/bench.py --no-cpu-baseline --configs main --steps 20 --warmup 5 --rows-file /tmp/rows.json > $OUT/prof_driver.log 2>&1
f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_coloring100k_f64_driver_cmd.csv && cut -c1-200 $OUT/kernel_stats_coloring100k_f64_driver_cmd.csv | head -3
rm -rf $OUT/p
if [ "$2" = "full" ]; then
echo "== kernel traces of the round's new rows (serial launches)"
for spec in coloring_100k_scalefree:f64 coloring_100k_scalefree:f32 coloring_1m_scalefree:f64 secp_100k:f64 secp_100k:f32 secp_100k_m4:f64 secp_100k_m4:f32 peav_50k:f64 secp_30k_m5:f64 meeting_5k_d40:f64; do
  IFS=: read wl dt <<< "$spec"
  rm -rf $OUT/p
  MAXSUM_NARY_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python3 $R/bench.py --workload $wl --dtype $dt --steps 100 --warmup 10 --no-cpu-baseline --rows-file /tmp/rows.json > $OUT/prof_${wl}_$dt.log 2>&1
  f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_${wl}_${dt}.csv && cut -c1-160 $OUT/kernel_stats_${wl}_${dt}.csv | sed -n 2,3p
  rm -rf $OUT/p
done
echo "== PMC traffic of the SECP rows on the closing code"
cd $R
bash scripts/gpu_pmc.sh $TAG/pmc "FETCH_SIZE WRITE_SIZE" "secp_100k:f64:0 secp_100k:f32:0 secp_100k_m4:f64:0 secp_100k_m4:f32:0 meeting_5k_d40:f64:0" > $OUT/pmc.log 2>&1
echo "== the stated multi-GPU prediction, re-measured on this code"
timeout 900 python3 tools/scale_prediction.py --out $OUT/scale_prediction.json > $OUT/scale_prediction.log 2>&1
grep '^{"n"' $OUT/scale_prediction.log | python3 -c "
import sys, json
for line in sys.stdin:
    d = json.loads(line); print('N', d['n'], 'compute', round(d['shard_compute_us'],1), 'loopback', round(d.get('shard_cycle_us_rccl_loopback',-1),1), d.get('predicted_speedup_vs_one_gpu'))"
fi
exit 0
