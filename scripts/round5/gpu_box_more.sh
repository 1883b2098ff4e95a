#!/bin/bash
# Round 5: the box kernel's new cases on the GPU -- int16 records in two passes, lane grids that overhang the table --
# parity, then against the lane-packed kernel (layout flag 32768) at configs[4]'s size.
TAG=${1:-r5_box_more}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bit_exact_vs_oracle and (box or nary or meeting)" ) 2>&1 | tail -6 | tee $OUT/pytest.txt
for w in meeting_50k_i16 meeting_50k_hetero meeting_50k; do for dt in f64 f32; do for fl in 0 32768; do
  timeout 400 python bench.py --workload $w --dtype $dt --configs main --no-cpu-baseline --steps 100 --warmup 10 --layout-flags $fl > $OUT/b.json 2> $OUT/b.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$w $dt flags=$fl", round(d["ms_per_step"]*1e3,1),"us  stored",round(r.get("frac_of_stored_bytes",0),3), d["config"]["factor_kernels"], r.get("table_storage"))
except Exception as e:
    print("FAILED $w $dt $fl", e); print(open("$OUT/b.err").read()[-400:])
PY
done; done; done 2>&1 | tee $OUT/ab.txt
exit 0
