#!/bin/bash
# amaxsum after the slot-word rewrite: the GPU parity tests, throughput (cold + warm), per-dispatch trace
TAG=${1:-r4_amx2}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_amaxsum.py -q -m gpu -x 2>&1 | tail -3 | tee $OUT/tests.txt
timeout 300 python tools/amaxsum_bench.py --no-oracle 10000 100000 2>&1 | grep '^{' | tee $OUT/bench.jsonl
bash scripts/round4/gpu_r4_amx_trace.sh $TAG 2>&1 | head -70
