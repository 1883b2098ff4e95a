#!/bin/bash
# round 2, call A: full-size parity tests (BASELINE configs 2-4) + the all-configs bench line
TAG=${1:-r02a}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
nproc; free -g | head -2
echo "== pytest full size"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size_bit_exact or north_star or config2" --durations=10 2>&1 | tail -25 | tee $OUT/pytest_full_size.txt
echo "== bench default (all configs)"
( time timeout 1200 python bench.py ) 2>&1 | tail -6 | tee $OUT/bench_default.json
