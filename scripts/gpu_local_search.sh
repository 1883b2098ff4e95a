#!/bin/bash
# DSA / MGM: GPU parity tests + cycles per second (slot kernels vs CSR walks)
TAG=${1:-local_search}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_dsa.py tests/test_gpu_mgm.py -x -q -m gpu --durations=4 2>&1 | tail -12 | tee $OUT/pytest.txt
timeout 300 python tools/local_search_bench.py | tee $OUT/local_search_bench.jsonl
