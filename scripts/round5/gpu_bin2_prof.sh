#!/bin/bash
# Round 5, call 3: where the time of the lane-grid factor kernels and of k_variable_wide goes on peav_50k /
# coloring_100k_d8 -- kernel traces WITHOUT the two-stream overlap, then PMC passes (one counter each).
TAG=${1:-r5_bin2_prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for w in peav_50k coloring_100k_d8; do
  for dt in f64 f32; do
  rm -rf /tmp/prof_$w
  MAXSUM_NARY_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o t -- python $R/bench.py --workload $w --dtype $dt \
      --configs main --no-cpu-baseline --steps 200 --warmup 20 > /tmp/prof_$w.log 2>&1
  f=$(find /tmp/prof_$w -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats_serial_${w}_${dt}.csv; echo "-- $w $dt (one stream)"; head -8 "$f" | cut -c1-200; fi
  done
done
MAXSUM_NARY_OVERLAP=0 bash scripts/gpu_pmc.sh $TAG/pmc "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY FETCH_SIZE WRITE_SIZE TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "peav_50k:f64:0 coloring_100k_d8:f64:0"
exit 0
