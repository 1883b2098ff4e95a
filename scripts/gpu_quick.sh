#!/bin/bash
# ad-hoc GPU call: parity tests + selected bench variants
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-quick}; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
: > $OUT/bench_variants.jsonl
free -g | head -2
for v in "" "--workload meeting_50k --steps 20 --warmup 3" "--workload meeting_50k --steps 20 --warmup 3 --dtype f32" "--workload meeting_50k --steps 20 --warmup 3 --layout-flags 16"; do
  echo "-- $v"
  (echo -n "{\"args\": \"$v\", \"out\": "; timeout 1200 python bench.py --no-cpu-baseline $v 2>&1 | tail -1; echo "}") | tee -a $OUT/bench_variants.jsonl | cut -c1-300
done
