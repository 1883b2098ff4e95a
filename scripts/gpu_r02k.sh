#!/bin/bash
# round 2, call K: side-stream overlap of the wide variable kernels with the n-ary launches
TAG=${1:-r02k}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for ov in 1 0; do
  export MAXSUM_NARY_OVERLAP=$ov
  for w in "meeting_50k --steps 100 --warmup 10" "meeting_50k --dtype f32 --steps 100 --warmup 10"; do
    timeout 300 python bench.py --no-cpu-baseline --configs main --workload $w 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('overlap=%s %-52s %9.2f us  frac %.3f  wall ms/step %.4f' % ('$ov', '$w', r['avg_launch_us'], r['frac'], d['ms_per_step']))" | tee -a $OUT/overlap_ab.txt
  done
done
unset MAXSUM_NARY_OVERLAP
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nary or meeting or wide or table_updates or hub" 2>&1 | tail -4 | tee $OUT/pytest.txt
