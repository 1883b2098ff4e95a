#!/bin/bash
# round 2, call J: prefetch depth of the packed n-ary kernel
TAG=${1:-r02j}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for lib in libmaxsum_hip.so libmaxsum_hip_pf2.so libmaxsum_hip_pf6.so libmaxsum_hip_pf8.so; do
  export MAXSUM_HIP_LIB=$R/pydcop_amd/csrc/$lib
  for w in "meeting_50k --steps 100 --warmup 10" "meeting_50k --dtype f32 --steps 100 --warmup 10"; do
    timeout 300 python bench.py --no-cpu-baseline --configs main --workload $w 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-24s %-52s %9.2f us  frac %.3f' % ('$lib', '$w', r['avg_launch_us'], r['frac']))" | tee -a $OUT/pf_ab.txt
  done
done
