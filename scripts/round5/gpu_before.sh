#!/bin/bash
# Round 5, step 0: what the CLOSING CODE OF ROUND 4 does on the instances the round-4 verdict named
# (binary factors over 5 <= D <= 63 -> factor_generic).  Run through gpurun; writes gpurun_out/r5_before/.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r5_before
mkdir -p $OUT
export TMPDIR=/tmp
for w in peav_50k coloring_100k_d8; do
  for dt in f64 f32; do
    timeout 600 python bench.py --workload $w --dtype $dt --configs main --no-cpu-baseline --steps 20 --warmup 3 \
        > $OUT/bench_${w}_${dt}.json 2> $OUT/bench_${w}_${dt}.err
    tail -c 600 $OUT/bench_${w}_${dt}.json
  done
done
for w in peav_50k coloring_100k_d8; do
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o t -- python $OLDPWD/bench.py --workload $w --dtype f64 \
      --configs main --no-cpu-baseline --steps 20 --warmup 3 > /dev/null 2> /tmp/prof_$w.err)
  f=$(find /tmp/prof_$w -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $OUT/kernel_stats_${w}_f64.csv && head -8 "$f"
done
