#!/bin/bash
# round 2, call G: amaxsum on the GPU (parity + goldens + throughput), meeting_50k back to the r02a kernel
TAG=${1:-r02g}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== pytest amaxsum"
timeout 900 python -m pytest tests/test_gpu_amaxsum.py -x -q -m gpu --durations=5 2>&1 | tail -25 | tee $OUT/pytest_amaxsum.txt
echo "== amaxsum throughput"
timeout 600 python tools/amaxsum_bench.py 10000 100000 2>&1 | tail -4 | tee $OUT/amaxsum_bench.jsonl
echo "== meeting_50k"
for w in "meeting_50k --steps 100 --warmup 10" "meeting_50k --dtype f32 --steps 100 --warmup 10"; do
    timeout 600 python bench.py --no-cpu-baseline --configs main --workload $w 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-60s %9.2f us  frac %.3f %s' % ('$w', r['avg_launch_us'], r['frac'], r.get('table_storage')))" | tee -a $OUT/meeting.txt
done
