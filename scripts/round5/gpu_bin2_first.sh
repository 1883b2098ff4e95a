#!/bin/bash
# Round 5, call 2: the lane-grid kernel (bin_box.h) on the GPU -- parity, then the two instances against the
# round-4 library (libmaxsum_hip_r4.so = factor_generic), then the kernel trace.
TAG=${1:-r5_bin2_first}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== parity: bin2 cases"
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bin2 or coloring5 or mixed or corner" ) 2>&1 | tail -6 | tee $OUT/pytest_bin2.txt
echo "== parity: full size"
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "full_size and (peav or d8)" --durations=6 ) 2>&1 | tail -12 | tee $OUT/pytest_full.txt
for w in peav_50k coloring_100k_d8; do
  for dt in f64 f32; do
    for lib in libmaxsum_hip.so; do
      [ -f pydcop_amd/csrc/$lib ] || continue
      MAXSUM_HIP_LIB=$R/pydcop_amd/csrc/$lib timeout 600 python bench.py --workload $w --dtype $dt --configs main --no-cpu-baseline --steps 200 --warmup 20 \
          > $OUT/bench_${w}_${dt}_${lib%.so}.json 2> $OUT/bench_${w}_${dt}_${lib%.so}.err
      python - <<PY
import json
d=json.loads(open("$OUT/bench_${w}_${dt}_${lib%.so}.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$w $dt $lib", round(d["ms_per_step"]*1e3,1),"us  event",round(r["avg_launch_us"],1),"frac",round(r["frac"],3),"stored",r.get("stored_bytes_per_launch"),round(r.get("frac_of_stored_bytes",0),3),r.get("table_storage"),"launches",r.get("launches_per_cycle"))
PY
    done
  done
done 2>&1 | tee $OUT/bench_summary.txt
for w in peav_50k coloring_100k_d8; do
  for dt in f64 f32; do
  rm -rf /tmp/prof_$w
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o t -- python $R/bench.py --workload $w --dtype $dt \
      --configs main --no-cpu-baseline --steps 200 --warmup 20 > /tmp/prof_$w.log 2>&1
  f=$(find /tmp/prof_$w -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats_${w}_${dt}.csv; echo "-- $w $dt"; head -8 "$f" | cut -c1-220; fi
  done
done
exit 0
