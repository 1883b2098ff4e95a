#!/bin/bash
# A/B timing of bench variants in one GPU call (no cpu baseline).
# usage: gpurun --timeout 600 -- 'bash scripts/gpu_ab.sh TAG [pytest-k-expression|-] "bench args 1" "bench args 2" ...'
TAG=${1:-ab}; shift
KEXPR=${1:--}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
if [ "$KEXPR" != "-" ]; then
  echo "== pytest -m gpu -k '$KEXPR'"
  timeout 900 python -m pytest tests -x -q -m gpu -k "$KEXPR" 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
fi
: > $OUT/bench_variants.jsonl
for v in "$@"; do
  (echo -n "{\"args\": \"$v\", \"out\": "; timeout 600 python bench.py --no-cpu-baseline $v 2>&1 | tail -1; echo "}") >> $OUT/bench_variants.jsonl
done
python scripts/parse_variants.py $OUT/bench_variants.jsonl
