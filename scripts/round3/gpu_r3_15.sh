#!/bin/bash
TAG=${1:-r3_15}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
bash scripts/gpu_ab_lib.sh $TAG/pl "libmaxsum_hip.so libmaxsum_hip_pl.so libmaxsum_hip.so libmaxsum_hip_pl.so" "--configs main --steps 3000 --warmup 300" "--configs main --dtype f32 --steps 3000 --warmup 300" "--configs main --workload ising_1024 --steps 500 --warmup 50" "--configs main --workload coloring_1m_deg6 --steps 300 --warmup 30" "--configs main --workload coloring_10k --steps 4000 --warmup 400" 2>&1 | tee $OUT/preload_ab.txt
