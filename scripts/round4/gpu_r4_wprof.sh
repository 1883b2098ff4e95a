#!/bin/bash
# phase clocks of k_variable_wide (profiling build -DMXS_WIDE_PROFILE), serial launches
TAG=${1:-r4_wprof}; LIBS=${2:-libmaxsum_hip_wprof.so}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for l in $LIBS; do for dt in f64; do
  echo "== $l $dt"
  MAXSUM_HIP_LIB=$R/pydcop_amd/csrc/$l MAXSUM_NARY_OVERLAP=0 timeout 300 python bench.py --no-cpu-baseline --configs main --workload meeting_50k --dtype $dt --steps 100 --warmup 10 2>&1 | grep -E "phase clocks|ms_per_step" | cut -c1-400 | tee -a $OUT/wprof.txt
done; done
