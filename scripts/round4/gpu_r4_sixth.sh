#!/bin/bash
# round 4, call 6: evidence batch -- non-temporal stores on tight records A/B, rocprofv3 kernel traces of every
# BASELINE configuration the bench line quotes, PMC traffic of the instances whose kernels changed, amaxsum trace.
TAG=${1:-r4_sixth}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== nt stores on D = 3 records: default vs -DMXS_NT_TIGHT=0"
bash scripts/gpu_ab_lib.sh $TAG/nt "libmaxsum_hip.so libmaxsum_hip_ntoff.so" \
  "--configs main --workload coloring_1m_deg6 --dtype f64 --steps 400 --warmup 40" \
  "--configs main --workload coloring_1m_deg6 --dtype f32 --steps 400 --warmup 40" \
  "--configs main --workload coloring_100k --dtype f64 --steps 2000 --warmup 200" 2>&1 | tail -8
echo "== kernel traces"
cd /tmp
for spec in coloring_100k:f64:2000 coloring_100k:f32:2000 coloring_10k:f64:2000 ising_1024:f64:400 ising_1024:f32:400 coloring_1m_deg6:f64:300 coloring_1m_deg6:f32:300 meeting_50k:f64:200 meeting_50k:f32:200; do
  IFS=: read w dt st <<< "$spec"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python $R/bench.py --no-cpu-baseline --configs main --workload $w --dtype $dt --steps $st --warmup $((st/10)) > $OUT/prof_${w}_${dt}.log 2>&1
  f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats_${w}_${dt}.csv; echo "$w $dt: $(sed -n 2p $f | cut -c1-150)"; tail -1 $OUT/prof_${w}_${dt}.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('   bench line of the traced run: ms_per_step', d['ms_per_step'], 'avg_launch_us', r['avg_launch_us'], 'frac', r['frac'])"; fi
  rm -rf $OUT/p
done
cd $R
echo "== PMC traffic"
bash scripts/gpu_pmc.sh $TAG/pmc "FETCH_SIZE WRITE_SIZE" "meeting_50k:f64:0 meeting_50k:f32:0 coloring_1m_deg6:f64:0 coloring_1m_deg6:f32:0" 2>&1 | grep -v "^  (" | cut -c1-150 | tail -24
echo "== amaxsum"
bash scripts/gpu_amaxsum.sh $TAG/amaxsum 2>&1 | tail -24
