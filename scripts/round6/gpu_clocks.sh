#!/bin/bash
# Round 6: which engine clock do the short latency-bound kernels run at?  rocm-smi sampled while bench.py loops.
TAG=${1:-r6_clocks}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for wl in coloring_100k_scalefree coloring_100k meeting_50k; do
  mode=""
  timeout 120 python3 bench.py --workload $wl --steps 20000 --warmup 100 --no-cpu-baseline --rows-file /tmp/r.json > $OUT/bench_$wl.json 2>/dev/null &
  pid=$!
  sleep 14
  for i in 1 2 3 4 5; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | tr -s ' ' | head -4; echo --; sleep 0.7; done > $OUT/clocks_$wl.txt
  wait $pid
  echo "== $wl"; sort $OUT/clocks_$wl.txt | uniq -c | sort -rn | head -8
  python3 -c "
import json; d=json.loads(open('$OUT/bench_$wl.json').read().strip().splitlines()[-1]); print('$wl', round(d['ms_per_step']*1e3,2), 'us')"
done
rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -3
exit 0
