#!/bin/bash
# the full GPU suite, smoke, the bench line, kernel trace of the bench command
TAG=${1:-validate}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== pytest -m gpu (everything)"
( time timeout 1100 python -m pytest tests -x -q -m gpu --durations=12 ) 2>&1 | tail -32 | tee $OUT/pytest_gpu.txt
echo "== smoke"
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -4 | tee $OUT/smoke.txt
echo "== bench default"
( time timeout 900 python bench.py ) 2>&1 | tail -5 | tee $OUT/bench_default.json
echo "== rocprofv3 kernel trace of the metric's configuration"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python $R/bench.py --no-cpu-baseline --configs main --steps 2000 --warmup 200 > $OUT/prof.log 2>&1
f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_coloring100k.csv && cut -c1-200 $OUT/kernel_stats_coloring100k.csv | head -6; rm -rf $OUT/p
tail -1 $OUT/prof.log | cut -c1-300
