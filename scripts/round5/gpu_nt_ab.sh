#!/bin/bash
# Round 5: non-temporal loads of the lane-grid table images (MAXSUM_STREAMING=1/0) on peav_50k; k_variable_pack8 with
# the one-domain-size fast path on coloring_100k_d8.
TAG=${1:-r5_nt_ab}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for rep in 1 2; do for dt in f64 f32; do for st in 1 0; do
  MAXSUM_STREAMING=$st timeout 300 python bench.py --workload peav_50k --dtype $dt --configs main --no-cpu-baseline --steps 200 --warmup 20 > $OUT/b.json 2> $OUT/b.err
  python - <<PY
import json
d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("peav_50k $dt nt_tables=$st", round(d["ms_per_step"]*1e3,1),"us  frac",round(r["frac"],3),"stored",round(r.get("frac_of_stored_bytes",0),3))
PY
done; done; done 2>&1 | tee $OUT/ab_nt.txt
for dt in f64 f32; do
  timeout 300 python bench.py --workload coloring_100k_d8 --dtype $dt --configs main --no-cpu-baseline --steps 200 --warmup 20 > $OUT/b.json 2> $OUT/b.err
  python - <<PY
import json
d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("coloring_100k_d8 $dt", round(d["ms_per_step"]*1e3,1),"us  frac",round(r["frac"],3),"stored",round(r.get("frac_of_stored_bytes",0),3))
PY
done 2>&1 | tee $OUT/d8.txt
exit 0
