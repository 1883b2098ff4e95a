#!/bin/bash
# round 2, call D: compact-table A/B + PMC, new GPU tests (dynamic, plug-in, layout variants)
TAG=${1:-r02d}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== A/B compact tables (0 = on, 8192 = full width)"
: > $OUT/ab.txt
for w in "coloring_100k --steps 2000 --warmup 200" "coloring_100k --dtype f32 --steps 2000 --warmup 200" \
         "coloring_10k --steps 2000 --warmup 200" "coloring_100k_hard --steps 2000 --warmup 200" \
         "coloring_1m_deg6 --steps 200 --warmup 20" "coloring_1m_deg6 --dtype f32 --steps 200 --warmup 20"; do
  for f in 0 8192; do
    timeout 300 python bench.py --no-cpu-baseline --configs main --workload $w --layout-flags $f 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-60s flags %5d  %9.2f us  frac %.3f' % ('$w', $f, d['roofline']['avg_launch_us'], d['roofline']['frac']))" | tee -a $OUT/ab.txt
  done
done
echo "== pytest new GPU tests"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plugin.py -x -q -m gpu -k "dynamic or plugin or proxies or layout_variants or table_updates or native_library or tables_ or coloring3_soft" 2>&1 | tail -8 | tee $OUT/pytest_new.txt
echo "== PMC"
bash scripts/gpu_pmc2.sh $TAG "FETCH_SIZE WRITE_SIZE" "coloring_100k:f64:0 coloring_1m_deg6:f64:0 coloring_100k:f32:0 coloring_1m_deg6:f32:0" 2>&1 | grep -v "^  ("
