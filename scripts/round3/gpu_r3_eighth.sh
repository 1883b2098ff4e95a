#!/bin/bash
TAG=${1:-r3_eighth}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
bash scripts/gpu_wide_phases.sh $TAG/phases "libmaxsum_hip.so"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "meeting or wide or corner or mixed or hub" 2>&1 | tail -2
for dt in f64 f32; do echo -n "meeting $dt: "; timeout 300 python bench.py --no-cpu-baseline --configs main --workload meeting_50k --dtype $dt --steps 300 --warmup 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1),'us')"; done
