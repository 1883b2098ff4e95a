#!/bin/bash
# round 2, call I: lane-packed narrow n-ary tables: A/B on meeting_50k, parity, kernel trace
TAG=${1:-r02i}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for w in "meeting_50k --steps 100 --warmup 10" "meeting_50k --dtype f32 --steps 100 --warmup 10"; do
  for f in 0 8192; do
    timeout 300 python bench.py --no-cpu-baseline --configs main --workload $w --layout-flags $f 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-55s flags %5d  %9.2f us  frac %.3f  stored-frac %.3f %s' % ('$w', $f, r['avg_launch_us'], r['frac'], r.get('frac_of_stored_bytes', 0), r.get('table_storage')))" | tee -a $OUT/ab.txt
  done
done
echo "== pytest"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nary or meeting or table_updates or dynamic or layout_variants" 2>&1 | tail -5 | tee $OUT/pytest.txt
echo "== kernel trace meeting_50k"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python $R/bench.py --no-cpu-baseline --configs main --workload meeting_50k --steps 50 --warmup 5 > $OUT/prof.log 2>&1
f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_meeting50k.csv && cut -c1-200 $OUT/kernel_stats_meeting50k.csv | head -5; rm -rf $OUT/p
