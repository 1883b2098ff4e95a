R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_tail; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
( timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "test_bit_exact_vs_oracle or secp_30k_m5 or table_updates" 2>&1 | tail -3 ) | tee $OUT/parity.txt
for spec in secp_30k_m5:f64 secp_30k_m5:f32 meeting_5k_d40:f64 meeting_50k_float:f64; do
  IFS=: read wl dt <<< "$spec"
  timeout 400 python3 bench.py --workload $wl --dtype $dt --steps 50 --warmup 5 --no-cpu-baseline --rows-file /tmp/rows.json 2>&1 | tail -1 > $OUT/b.json
  python3 -c "
import json; d=json.loads(open('$OUT/b.json').read()); print('$wl $dt', round(d['ms_per_step']*1e3,2), 'us/cycle  min', round(d['timing']['ms_per_step_min']*1e3,2))" 2>&1 | tail -1
done | tee $OUT/ab.txt
