#!/bin/bash
TAG=${1:-r3_ninth}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
bash scripts/gpu_ab_lib.sh $TAG/waves "libmaxsum_hip_w4.so libmaxsum_hip.so" "--configs main --steps 3000 --warmup 300" "--configs main --dtype f32 --steps 3000 --warmup 300" "--configs main --workload coloring_1m_deg6 --steps 300 --warmup 30" "--configs main --workload coloring_1m_deg6 --dtype f32 --steps 300 --warmup 30" "--configs main --workload ising_1024 --steps 500 --warmup 50" "--configs main --workload ising_1024 --dtype f32 --steps 500 --warmup 50" "--configs main --workload coloring_10k --steps 4000 --warmup 400" 2>&1 | tee $OUT/waves_ab.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "north_star or config2 or (bit_exact_vs_oracle and not full_size)" 2>&1 | tail -2
