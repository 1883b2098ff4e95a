#!/bin/bash
# lean (scheduled) instantiation of the sweep, SGPR cap 80 vs 104
TAG=${1:-r3_12}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
bash scripts/gpu_ab_lib.sh $TAG/sgpr "libmaxsum_hip.so libmaxsum_hip_sg104.so" "--configs main --steps 3000 --warmup 300" "--configs main --dtype f32 --steps 3000 --warmup 300" "--configs main --workload coloring_10k --steps 4000 --warmup 400" "--configs main --workload coloring_10k --dtype f32 --steps 4000 --warmup 400" "--configs main --workload coloring_1m_deg6 --steps 300 --warmup 30" "--configs main --workload ising_1024 --steps 500 --warmup 50" 2>&1 | tee $OUT/sgpr_ab.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_sharded.py -x -q -m gpu -k "north_star or config2 or (bit_exact_vs_oracle and not full_size) or shard" 2>&1 | tail -2
