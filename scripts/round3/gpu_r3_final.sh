#!/bin/bash
# the round's closing call: full GPU suite + smoke + bench + kernel trace, PMC passes, meeting kernel times
TAG=${1:-r3_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
bash scripts/gpu_validate.sh $TAG/validate
bash scripts/gpu_wide_phases.sh $TAG/phases "libmaxsum_hip.so"
bash scripts/gpu_pmc.sh $TAG/pmc "FETCH_SIZE WRITE_SIZE" "coloring_100k:f64:0 coloring_100k:f32:0 coloring_1m_deg6:f64:0 coloring_1m_deg6:f32:0 ising_1024:f64:0 ising_1024:f32:0 meeting_50k:f64:0 meeting_50k:f32:0" 2>&1 | grep -v "^  (" | cut -c1-150 | tail -30
