#!/bin/bash
TAG=${1:-r3_14}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
bash scripts/gpu_ab_lib.sh $TAG/final "libmaxsum_hip.so" "--configs main --steps 3000 --warmup 300" "--configs main --dtype f32 --steps 3000 --warmup 300" "--configs main --workload ising_1024 --steps 500 --warmup 50" "--configs main --workload ising_1024 --dtype f32 --steps 500 --warmup 50" "--configs main --workload coloring_1m_deg6 --steps 300 --warmup 30" "--configs main --workload coloring_10k --steps 4000 --warmup 400" 2>&1 | tee $OUT/final_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_sharded.py -x -q -m gpu 2>&1 | tail -2
