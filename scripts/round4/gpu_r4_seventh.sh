#!/bin/bash
# round 4, call 7: DSA / MGM with the row view of the variables the pack cannot take (meeting_50k)
TAG=${1:-r4_seventh}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_dsa.py tests/test_gpu_mgm.py -x -q -m gpu --durations=4 2>&1 | tail -10 ) | tee $OUT/pytest.txt
timeout 600 python tools/local_search_bench.py --instances meeting_50k --kernels packed strided | tee $OUT/local_search_bench.jsonl
timeout 300 python tools/local_search_bench.py --instances coloring_100k --kernels packed | tee -a $OUT/local_search_bench.jsonl
