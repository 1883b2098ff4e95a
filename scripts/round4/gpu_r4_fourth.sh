#!/bin/bash
# round 4, call 4: k_variable_wide, one workgroup per block again, with the new chain loop / batched requests /
# LDS-only barriers; register budget for 8 waves; phase clocks
TAG=${1:-r4_fourth}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
bash scripts/gpu_wide_phases.sh $TAG/phases "libmaxsum_hip.so" 0 f64
bash scripts/gpu_wide_phases.sh $TAG/phases "libmaxsum_hip.so" 0 f32
bash scripts/round4/gpu_r4_wprof.sh $TAG/wprof libmaxsum_hip_wprof.so
for dt in f64 f32; do
  timeout 300 python bench.py --no-cpu-baseline --configs main --workload meeting_50k --dtype $dt --steps 300 --warmup 30 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$dt', d['ms_per_step']*1000, 'us/cycle')" | tee -a $OUT/cycle.txt
done
( time timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "bit_exact_vs_oracle and (wide or meeting or hub or mixed or nary)" ) 2>&1 | tail -4 | tee $OUT/pytest.txt
