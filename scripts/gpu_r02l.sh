#!/bin/bash
# round 2, call L: four-value butterfly reduction in the n-ary kernels (variant library) A/B + parity
TAG=${1:-r02l}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for lib in libmaxsum_hip.so libmaxsum_hip_r4.so; do
  export MAXSUM_HIP_LIB=$R/pydcop_amd/csrc/$lib
  for w in "meeting_50k --steps 100 --warmup 10" "meeting_50k --dtype f32 --steps 100 --warmup 10" "meeting_50k --steps 100 --warmup 10 --layout-flags 8192"; do
    timeout 300 python bench.py --no-cpu-baseline --configs main --workload $w 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-22s %-62s %9.2f us  frac %.3f' % ('$lib', '$w', r['avg_launch_us'], r['frac']))" | tee -a $OUT/r4_ab.txt
  done
done
MAXSUM_HIP_LIB=$R/pydcop_amd/csrc/libmaxsum_hip_r4.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nary or meeting or table_updates" 2>&1 | tail -4 | tee $OUT/pytest_r4.txt
