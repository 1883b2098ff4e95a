#!/bin/bash
# ad-hoc GPU call: pattern bench + runtime-sharing checks
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-quick}; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== torch sees the GPU?"; timeout 300 python -c "import torch; print(torch.cuda.is_available(), torch.cuda.device_count(), torch.version.hip)" 2>&1 | tail -2
echo "== pattern bench"; timeout 600 ./tools/pattern_bench 2>&1 | tee $OUT/pattern_bench.jsonl
echo "== smoke (system runtime)"; timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3
echo "== pytest gpu (torch runtime)"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
