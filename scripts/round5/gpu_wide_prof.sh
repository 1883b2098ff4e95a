#!/bin/bash
# Round 5: phase clocks of k_variable_wide (profiling build, -DMXS_WIDE_PROFILE) on the instances of this round;
# parity of the round's last kernel changes (box overhang, small-row n-ary).
TAG=${1:-r5_wide_prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for w in peav_50k meeting_50k coloring_100k_d8; do
  fl=0; [ $w = coloring_100k_d8 ] && fl=1048576
  MAXSUM_HIP_LIB=$R/pydcop_amd/csrc/libmaxsum_hip_wprof.so MAXSUM_NARY_OVERLAP=0 timeout 300 python bench.py --workload $w --dtype f64 --configs main --no-cpu-baseline --steps 100 --warmup 10 --layout-flags $fl 2>&1 >/dev/null | grep "phase clocks" | sed "s/^/$w: /"
done | tee $OUT/wide_phases.txt
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "(bit_exact_vs_oracle and not full_size) or table_updates" ) 2>&1 | tail -8 | tee $OUT/pytest_parity.txt
exit 0
