#!/bin/bash
TAG=${1:-r3_13}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
bash scripts/gpu_ab_lib.sh $TAG/sgpr "libmaxsum_hip.so libmaxsum_hip_sg104.so libmaxsum_hip_sg112.so libmaxsum_hip.so libmaxsum_hip_sg104.so libmaxsum_hip_sg112.so" "--configs main --steps 3000 --warmup 300" "--configs main --workload ising_1024 --steps 500 --warmup 50" "--configs main --workload ising_1024 --dtype f32 --steps 500 --warmup 50" "--configs main --workload coloring_1m_deg6 --dtype f32 --steps 300 --warmup 30" 2>&1 | tee $OUT/sgpr_ab.txt
