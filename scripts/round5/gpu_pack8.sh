#!/bin/bash
# Round 5: the lane-per-edge kernel of the variables of 5..8 values (k_variable_pack8) on the GPU: parity, A/B against the wide
# class (flag 1048576) on coloring_100k_d8, kernel traces of the rows the bench line gained this round, PMC of the full-width 24^3 path.
TAG=${1:-r5_pack8}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== parity"
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "(bit_exact_vs_oracle and not full_size) or table_updates or layout_variants" ) 2>&1 | tail -4 | tee $OUT/pytest_parity.txt
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "full_size and (d8 or peav)" ) 2>&1 | tail -4 | tee $OUT/pytest_full.txt
for dt in f64 f32; do for fl in 0 1048576; do
  timeout 600 python bench.py --workload coloring_100k_d8 --dtype $dt --configs main --no-cpu-baseline --steps 200 --warmup 20 --layout-flags $fl > $OUT/b.json 2> $OUT/b.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("coloring_100k_d8 $dt flags=$fl", round(d["ms_per_step"]*1e3,1),"us  frac",round(r["frac"],3),"stored",round(r.get("frac_of_stored_bytes",0),3),"launches",r.get("launches_per_cycle"))
except Exception as e:
    print("FAILED", e); print(open("$OUT/b.err").read()[-500:])
PY
done; done 2>&1 | tee $OUT/ab_pack8.txt
for spec in coloring_100k_d8:f64 coloring_100k_d8:f32 peav_50k:f64 peav_50k:f32 meeting_50k_float:f64 meeting_50k_float:f32 meeting_50k:f64 meeting_50k:f32 ising_1024:f64 ising_1024:f32; do
  IFS=: read w dt <<< "$spec"
  rm -rf /tmp/prof_t
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o t -- python $R/bench.py --workload $w --dtype $dt \
      --configs main --no-cpu-baseline --steps 200 --warmup 20 > /tmp/prof_t.log 2>&1
  f=$(find /tmp/prof_t -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats_${w}_${dt}.csv; echo "-- $w $dt"; head -6 "$f" | cut -d, -f1-4 | cut -c1-170; fi
done
bash scripts/gpu_pmc.sh $TAG/pmc "FETCH_SIZE WRITE_SIZE" "meeting_50k_float:f64:0 meeting_50k_float:f32:0 peav_50k:f32:0 coloring_100k_d8:f32:0 ising_1024:f64:0 ising_1024:f32:0"
exit 0
