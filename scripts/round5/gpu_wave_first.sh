#!/bin/bash
# Round 5, call 4: the one-wave-per-run variable kernel (k_variable_wave) on the GPU: parity, then A/B against the
# workgroup-per-run kernel (layout flag 1048576) on the instances with wide variables, overlap on / off.
TAG=${1:-r5_wave_first}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== parity"
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "bit_exact_vs_oracle and not full_size" ) 2>&1 | tail -5 | tee $OUT/pytest_parity.txt
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "full_size and (peav or d8 or meeting_50k-)" ) 2>&1 | tail -5 | tee $OUT/pytest_full.txt
for w in peav_50k coloring_100k_d8 meeting_50k; do
  for dt in f64 f32; do
    for fl in 0 1048576; do
      for ov in 1 0; do
      MAXSUM_NARY_OVERLAP=$ov timeout 600 python bench.py --workload $w --dtype $dt --configs main --no-cpu-baseline --steps 200 --warmup 20 --layout-flags $fl \
          > $OUT/b.json 2> $OUT/b.err
      python - <<PY
import json
try:
    d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("$w $dt flags=$fl overlap=$ov", round(d["ms_per_step"]*1e3,1),"us  frac",round(r["frac"],3),"stored",round(r.get("frac_of_stored_bytes",0),3),"launches",r.get("launches_per_cycle"), "reps", d["timing"]["repeats"])
except Exception as e:
    print("$w $dt flags=$fl overlap=$ov FAILED", e); print(open("$OUT/b.err").read()[-600:])
PY
      done
    done
  done
done 2>&1 | tee $OUT/ab_summary.txt
for w in peav_50k coloring_100k_d8; do
  rm -rf /tmp/prof_$w
  MAXSUM_NARY_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o t -- python $R/bench.py --workload $w --dtype f64 \
      --configs main --no-cpu-baseline --steps 200 --warmup 20 > /tmp/prof_$w.log 2>&1
  f=$(find /tmp/prof_$w -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats_serial_${w}_f64.csv; echo "-- $w f64 (one stream)"; head -8 "$f" | cut -d, -f1-4 | cut -c1-160; fi
done
exit 0
