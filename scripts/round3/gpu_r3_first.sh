#!/bin/bash
# round 3, first call: the reference on the GPU box (e2e CLI, reference-vs-HIP, goldens with
# messages), the bandwidth sweep, the default bench line (reference thread agents timed live)
TAG=${1:-r3_first}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
python oracle/stage_reference.py locate
echo "== new gpu tests"
( time timeout 900 python -m pytest tests/test_gpu_reference_e2e.py tests/test_gpu_vs_reference.py tests/test_gpu_plugin.py -x -q -m gpu --durations=8 ) 2>&1 | tail -25 | tee $OUT/pytest_new.txt
( time timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden" ) 2>&1 | tail -6 | tee $OUT/pytest_golden.txt
echo "== bw sweep"
timeout 300 tools/bw_sweep 1024 > $OUT/bw_sweep_1024.jsonl 2>&1; sort -t: -k9 $OUT/bw_sweep_1024.jsonl | tail -3
timeout 120 tools/bw_sweep 4096 > $OUT/bw_sweep_4096.jsonl 2>&1; tail -2 $OUT/bw_sweep_4096.jsonl
echo "== bench default"
( time timeout 900 python bench.py ) 2>&1 | tail -5 | tee $OUT/bench_default.json | cut -c1-1500
