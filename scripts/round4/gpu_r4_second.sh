#!/bin/bash
# round 4, call 2: the one-wave-per-factor box kernel (nary_box.h) -- parity, A/B against the lane-packed
# kernel (layout flag 32768), kernel times, VALU counters; valu_bench with the select variants.
TAG=${1:-r4_second}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== valu_bench"
timeout 120 tools/valu_bench 3000 > $OUT/valu_bench.jsonl 2>&1; grep -E "cndmask|select" $OUT/valu_bench.jsonl | cut -c1-160
echo "== parity (n-ary paths, fuzz, reference, table updates, dynamic)"
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_vs_reference.py -x -q -k "bit_exact_vs_oracle or table_updates or dynamic or fuzz or reference or golden" ) 2>&1 | tail -8 | tee $OUT/pytest.txt
echo "== meeting_50k A/B: box (flags 0) vs lane-packed (flags 32768)"
for fl in 0 32768; do for dt in f64 f32; do
  timeout 300 python bench.py --no-cpu-baseline --configs main --workload meeting_50k --dtype $dt --layout-flags $fl --steps 300 --warmup 30 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('flags=$fl $dt', d['ms_per_step']*1000, 'us/cycle', json.dumps(d.get('roofline'))[:300])" | tee -a $OUT/ab.txt
done; done
echo "== kernel times (serial launches)"
bash scripts/gpu_wide_phases.sh $TAG/phases "libmaxsum_hip.so" 0 f64
bash scripts/gpu_wide_phases.sh $TAG/phases "libmaxsum_hip.so" 0 f32
echo "== counters of the box kernel"
bash scripts/gpu_meeting_pmc.sh $TAG/pmc "SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" 0 f64 2>&1 | cut -c1-150
