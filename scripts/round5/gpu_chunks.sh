#!/bin/bash
# Round 5: k_factor_bin with a row's 16-byte chunks interleaved across the lanes -- parity, the two instances, serial trace.
TAG=${1:-r5_chunks}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "(bit_exact_vs_oracle and not full_size) or table_updates or (full_size and (peav or d8))" ) 2>&1 | tail -8 | tee $OUT/pytest_parity.txt
for rep in 1 2; do for w in peav_50k coloring_100k_d8; do for dt in f64 f32; do
  timeout 300 python bench.py --workload $w --dtype $dt --configs main --no-cpu-baseline --steps 200 --warmup 20 > $OUT/b.json 2> $OUT/b.err
  python - <<PY
import json
d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$w $dt", round(d["ms_per_step"]*1e3,1),"us  frac",round(r["frac"],3),"stored",round(r.get("frac_of_stored_bytes",0),3))
PY
done; done; done 2>&1 | tee $OUT/bench.txt
for dt in f64 f32; do
rm -rf /tmp/prof_t
MAXSUM_NARY_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o t -- python $R/bench.py --workload peav_50k --dtype $dt \
    --configs main --no-cpu-baseline --steps 200 --warmup 20 > /tmp/prof_t.log 2>&1
f=$(find /tmp/prof_t -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats_serial_peav_50k_$dt.csv; echo "-- peav $dt (one stream)"; head -8 "$f" | cut -d, -f1-4 | cut -c1-170; fi
done
bash scripts/gpu_pmc.sh $TAG/pmc "TCP_TOTAL_CACHE_ACCESSES_sum FETCH_SIZE" "peav_50k:f64:0"
exit 0
