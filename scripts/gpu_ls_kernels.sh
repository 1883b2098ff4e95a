#!/bin/bash
# per-kernel times of the DSA / MGM engines for a list of library builds: gpu_ls_kernels.sh TAG "lib1 lib2 ..."
TAG=${1:-ls_kernels}; LIBS=${2:-libmaxsum_hip.so}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for l in $LIBS; do
  export MAXSUM_HIP_LIB=$R/pydcop_amd/csrc/$l
  rm -rf $OUT/p
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python $R/tools/local_search_bench.py --cycles 200 --instances coloring_100k --kernels packed > $OUT/ls_$l.log 2>&1
  f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1)
  echo "== $l" | tee -a $OUT/kernels.txt
  [ -n "$f" ] && python -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'dsa' in r['Name'] or 'mgm' in r['Name']: print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1000,2), 'us')" $f | tee -a $OUT/kernels.txt
  grep us_per_cycle $OUT/ls_$l.log | tee -a $OUT/kernels.txt
done; rm -rf $OUT/p
