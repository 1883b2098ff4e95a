#!/bin/bash
# PMC passes (one counter per pass) over the packed DSA / MGM kernels: gpu_ls_pmc.sh TAG "COUNTER ..."
TAG=${1:-ls_pmc}; CTRS=${2:-"TCC_REQ_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for c in $CTRS; do
  timeout 200 rocprofv3 --pmc $c --output-format csv -d $OUT/p -o pmc -- python $R/tools/local_search_bench.py --cycles 20 --instances coloring_100k --kernels packed > $OUT/log_$c.txt 2>&1
  f=$(find $OUT/p -name "*counter_collection*.csv" | head -1)
  if [ -n "$f" ]; then python $R/scripts/pmc_summary.py "$f" | grep -E "dsa|mgm" | tee -a $OUT/pmc_local_search.txt; else echo "FAILED $c"; tail -3 $OUT/log_$c.txt; fi
  rm -rf $OUT/p
done
