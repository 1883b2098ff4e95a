#!/bin/bash
TAG=${1:-r3_tenth}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
bash scripts/gpu_wide_phases.sh $TAG/phases "libmaxsum_hip.so"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vs_reference.py -x -q -m gpu 2>&1 | tail -2
for dt in f64 f32; do echo -n "meeting $dt: "; timeout 300 python bench.py --no-cpu-baseline --configs main --workload meeting_50k --dtype $dt --steps 300 --warmup 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1),'us', d['roofline']['launches_per_cycle'])"; done
echo "== kernel trace of the local-search engines"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python $R/tools/local_search_bench.py --cycles 200 > $OUT/ls.log 2>&1
f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_local_search.csv && python -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'dsa' in r['Name'] or 'mgm' in r['Name']: print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1000,1), 'us')" $OUT/kernel_stats_local_search.csv; rm -rf $OUT/p
