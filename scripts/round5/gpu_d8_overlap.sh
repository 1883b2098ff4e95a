#!/bin/bash
# Round 5: coloring_100k_d8 with k_variable_pack8 on the side stream beside k_factor_bin (MAXSUM_NARY_OVERLAP=1) or behind it (=0).
TAG=${1:-r5_d8_overlap}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for rep in 1 2; do for dt in f64 f32; do for ov in 0; do for fl in 0 2097152; do
  MAXSUM_NARY_OVERLAP=$ov timeout 300 python bench.py --workload coloring_100k_d8 --dtype $dt --configs main --no-cpu-baseline --steps 500 --warmup 50 --layout-flags $fl > $OUT/b.json 2> $OUT/b.err
  python - <<PY
import json
d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("coloring_100k_d8 $dt overlap=$ov flags=$fl", round(d["ms_per_step"]*1e3,1),"us  frac",round(r["frac"],3),"stored",round(r.get("frac_of_stored_bytes",0),3))
PY
done; done; done; done 2>&1 | tee $OUT/ab.txt
exit 0
