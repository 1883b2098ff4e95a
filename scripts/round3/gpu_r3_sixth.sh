#!/bin/bash
TAG=${1:-r3_sixth}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== DSA / MGM parity (packed, slot, CSR-walk kernels; 100k)"
( time timeout 900 python -m pytest tests/test_gpu_dsa.py tests/test_gpu_mgm.py tests/test_gpu_fuzz.py -x -q -m gpu ) 2>&1 | tail -5 | tee $OUT/pytest_ls.txt
echo "== local search bench: packed (default) / slot / CSR-walk kernels"
timeout 600 python tools/local_search_bench.py 2>&1 | tail -12 | tee $OUT/local_search_bench.jsonl
bash scripts/gpu_wide_phases.sh $TAG/phases "libmaxsum_hip.so"
echo "== bench default"
( time timeout 900 python bench.py ) 2>&1 | tail -5 | tee $OUT/bench_default.json | cut -c1-600
