#!/bin/bash
# Round 6, first call: (1) the DRIVER's exact command, raw stdout kept (the final line has to parse from the
# tail); (2) the BEFORE numbers of the hub-variable workloads on the round-5 kernels (variable_generic: one thread per hub).
TAG=${1:-r6_first}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== the driver's command, verbatim"
( time timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_stdout.txt 2> $OUT/bench_driver_stderr.txt ) 2>&1 | tail -3
tail -c 8192 $OUT/bench_driver_stdout.txt > $OUT/bench_driver_tail8k.txt
python3 - <<PY
import json
lines=open("$OUT/bench_driver_stdout.txt").read().strip().splitlines()
print("lines", len(lines), "last line bytes", len(lines[-1]))
d=json.loads(lines[-1]); print({k: d[k] for k in ("value","ms_per_step","n_gpus","steps","warmup")}, d["roofline"]["frac"], d["cpu_baseline"]["value"])
print(json.dumps(d["rows"]))
PY
cp bench_rows.json $OUT/bench_rows.json 2>/dev/null
echo "== BEFORE: scale-free colourings on the round-5 kernels"
for wl in coloring_100k_scalefree coloring_1m_scalefree; do
  for dt in f64 f32; do
    timeout 600 python3 bench.py --workload $wl --dtype $dt --steps 50 --warmup 5 --no-cpu-baseline --rows-file $OUT/rows_$wl_$dt.json 2>&1 | tail -1 > $OUT/before_${wl}_${dt}.json
    python3 -c "
import json; d=json.loads(open('$OUT/before_${wl}_${dt}.json').read()); print('$wl $dt', round(d['ms_per_step']*1e3,1), 'us/cycle, frac', round(d['roofline']['frac'],4), d['timing'])"
  done
done
echo "== rocprofv3 kernel trace, coloring_100k_scalefree f64 (before)"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python3 $R/bench.py --workload coloring_100k_scalefree --steps 50 --warmup 5 --no-cpu-baseline --rows-file /tmp/rows.json > $OUT/prof_sf.log 2>&1
f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_coloring_100k_scalefree_f64_before.csv && cut -c1-220 $OUT/kernel_stats_coloring_100k_scalefree_f64_before.csv | head -5
rm -rf $OUT/p
exit 0
