#!/bin/bash
# A/B between library builds: gpu_ab_lib.sh TAG "lib1 lib2 ..." "bench args 1" "bench args 2" ...
TAG=${1:-abl}; shift; LIBS=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
: > $OUT/bench_variants.jsonl
for v in "$@"; do for l in $LIBS; do
  export MAXSUM_HIP_LIB=$R/pydcop_amd/csrc/$l
  (echo -n "{\"args\": \"$l $v\", \"out\": "; timeout 600 python bench.py --no-cpu-baseline $v 2>&1 | tail -1; echo "}") >> $OUT/bench_variants.jsonl
done; done
python scripts/parse_variants.py $OUT/bench_variants.jsonl
