#!/bin/bash
TAG=${1:-wide_pmc}; CTRS=${2:-"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"}; FLAGS=${3:-0}; DT=${4:-f64}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for c in $CTRS; do
  MAXSUM_NARY_OVERLAP=0 timeout 200 rocprofv3 --pmc $c --output-format csv -d $OUT/p -o pmc -- python $R/bench.py --no-cpu-baseline --configs main --workload meeting_50k --dtype $DT --layout-flags $FLAGS --steps 10 --warmup 3 > $OUT/log_$c.txt 2>&1
  f=$(find $OUT/p -name "*counter_collection*.csv" | head -1)
  if [ -n "$f" ]; then python $R/scripts/pmc_summary.py "$f" | grep -E "k_variable_wide|k_factor_nary|k_factor_box" | tee -a $OUT/pmc_meeting_kernels.txt; else echo "FAILED $c"; tail -2 $OUT/log_$c.txt; fi
  rm -rf $OUT/p
done
