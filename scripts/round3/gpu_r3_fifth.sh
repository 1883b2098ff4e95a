#!/bin/bash
TAG=${1:-r3_fifth}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
bash scripts/gpu_wide_phases.sh $TAG/phases "libmaxsum_hip.so"
echo "== unpadded D = 3 records"
bash scripts/gpu_ab_lib.sh $TAG/tight "libmaxsum_hip.so libmaxsum_hip_tight.so" "--configs main --steps 3000 --warmup 300" "--configs main --dtype f32 --steps 3000 --warmup 300" "--configs main --workload coloring_1m_deg6 --steps 300 --warmup 30" "--configs main --workload coloring_1m_deg6 --dtype f32 --steps 300 --warmup 30" "--configs main --workload coloring_10k --steps 4000 --warmup 400" "--configs main --workload coloring_100k_hard --steps 3000 --warmup 300" 2>&1 | tee $OUT/tight_ab.txt
echo "== parity with the unpadded records"
MAXSUM_HIP_LIB=$R/pydcop_amd/csrc/libmaxsum_hip_tight.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bit_exact_vs_oracle and not full_size or north_star or config2" 2>&1 | tail -3 | tee $OUT/pytest_tight.txt
