#!/bin/bash
# round 4, call 3: the persistent, pipelined k_variable_wide (+ grid-size variants) and the box kernel with
# its message loads ahead of the table loads -- parity, kernel times, the meeting_50k cycle.
TAG=${1:-r4_third}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== parity (oracle cases, fuzz, full-size meeting)"
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "bit_exact_vs_oracle or fuzz" ) 2>&1 | tail -6 | tee $OUT/pytest.txt
echo "== kernel times (serial launches)"
bash scripts/gpu_wide_phases.sh $TAG/phases "libmaxsum_hip.so libmaxsum_hip_wg256.so libmaxsum_hip_wg512.so libmaxsum_hip_wg1024.so" 0 f64
bash scripts/gpu_wide_phases.sh $TAG/phases "libmaxsum_hip.so" 0 f32
echo "== meeting_50k cycle"
for dt in f64 f32; do
  timeout 300 python bench.py --no-cpu-baseline --configs main --workload meeting_50k --dtype $dt --steps 300 --warmup 30 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$dt', d['ms_per_step']*1000, 'us/cycle')" | tee -a $OUT/cycle.txt
done
