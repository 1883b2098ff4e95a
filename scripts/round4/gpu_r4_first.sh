#!/bin/bash
# round 4, call 1: instruction costs (tools/valu_bench), the new hard-constraint (+-inf / NaN) parity
# cases on the LDS kernels, the lengthened full-size comparisons, the reference-vs-HIP hard cases,
# and the meeting_50k kernel times before this round's kernel work.
TAG=${1:-r4_first}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== valu_bench"
timeout 120 tools/valu_bench 3000 > $OUT/valu_bench.jsonl 2>&1; tail -3 $OUT/valu_bench.jsonl | cut -c1-200
echo "== hard-constraint parity cases (oracle, reference)"
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vs_reference.py -x -q -k "hard_" ) 2>&1 | tail -8 | tee $OUT/pytest_hard.txt
echo "== full-size comparisons, long runs"
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "full_size_bit_exact" --durations=8 ) 2>&1 | tail -16 | tee $OUT/pytest_full.txt
echo "== meeting_50k kernel times"
bash scripts/gpu_wide_phases.sh $TAG/phases "libmaxsum_hip.so"
