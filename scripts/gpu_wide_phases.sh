#!/bin/bash
# where does k_variable_wide spend its time: phase-skip builds (make variant DEFS=-DMXS_WIDE_SKIP=..)
TAG=${1:-wide_phases}; FLAGS=${3:-0}; DT=${4:-f64}; LIBS=${2:-"libmaxsum_hip.so libmaxsum_hip_ws1.so libmaxsum_hip_ws2.so libmaxsum_hip_ws4.so libmaxsum_hip_ws8.so libmaxsum_hip_ws16.so libmaxsum_hip_ws31.so"}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for l in $LIBS; do
  MAXSUM_HIP_LIB=$R/pydcop_amd/csrc/$l MAXSUM_NARY_OVERLAP=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python $R/bench.py --no-cpu-baseline --configs main --workload meeting_50k --dtype $DT --layout-flags $FLAGS --steps 100 --warmup 10 > $OUT/prof.log 2>&1
  f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1)
  echo -n "$l flags=$FLAGS $DT: "; python -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_variable_wide' in r['Name'] or 'k_factor_nary' in r['Name'] or 'k_factor_box' in r['Name']: print(r['Name'].split('<')[0].split('::')[-1], round(float(r['AverageNs'])/1000,1), 'us', end='; ')
print()" "$f"
  rm -rf $OUT/p
done | tee -a $OUT/wide_phases.txt
