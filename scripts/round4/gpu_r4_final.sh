#!/bin/bash
# round 4 closing call: the whole -m gpu suite, smoke, the one-line bench, kernel trace of the metric's configuration,
# the meeting_50k kernel times and the box kernel's VALU counters on the final code
TAG=${1:-r4_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
bash scripts/gpu_validate.sh $TAG/validate
bash scripts/gpu_wide_phases.sh $TAG/phases "libmaxsum_hip.so" 0 f64
bash scripts/gpu_wide_phases.sh $TAG/phases "libmaxsum_hip.so" 0 f32
bash scripts/gpu_meeting_pmc.sh $TAG/pmc "SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" 0 f64 2>&1 | cut -c1-150
