#!/bin/bash
# round 2, call H: DPP wave-min variant of the n-ary kernel (A/B + parity), amaxsum tests (bounded)
TAG=${1:-r02h}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== meeting_50k: default vs DPP reductions"
for lib in libmaxsum_hip.so libmaxsum_hip_dpp.so; do
  export MAXSUM_HIP_LIB=$R/pydcop_amd/csrc/$lib
  for w in "meeting_50k --steps 100 --warmup 10" "meeting_50k --dtype f32 --steps 100 --warmup 10"; do
    timeout 300 python bench.py --no-cpu-baseline --configs main --workload $w 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-22s %-52s %9.2f us  frac %.3f' % ('$lib', '$w', r['avg_launch_us'], r['frac']))" | tee -a $OUT/dpp_ab.txt
  done
done
echo "== parity with the DPP library (n-ary cases + full-size meeting)"
MAXSUM_HIP_LIB=$R/pydcop_amd/csrc/libmaxsum_hip_dpp.so timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nary or meeting" 2>&1 | tail -4 | tee $OUT/pytest_dpp.txt
unset MAXSUM_HIP_LIB
echo "== pytest amaxsum + plugin"
timeout 400 python -m pytest tests/test_gpu_amaxsum.py tests/test_gpu_plugin.py -x -q -m gpu --durations=6 2>&1 | tail -14 | tee $OUT/pytest_amaxsum.txt
