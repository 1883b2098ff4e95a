#!/bin/bash
# Round 5: k_variable_wide with 512 threads per workgroup (twice the slots per chain phase) against 256.
TAG=${1:-r5_tpb}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for w in peav_50k meeting_50k meeting_50k_float; do for dt in f64 f32; do for lib in libmaxsum_hip.so libmaxsum_hip_tpb512.so; do for ov in 0 1; do
  MAXSUM_HIP_LIB=$R/pydcop_amd/csrc/$lib MAXSUM_NARY_OVERLAP=$ov timeout 300 python bench.py --workload $w --dtype $dt --configs main --no-cpu-baseline --steps 200 --warmup 20 > $OUT/b.json 2> $OUT/b.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$w $dt $lib overlap=$ov", round(d["ms_per_step"]*1e3,1),"us  frac",round(r["frac"],3),"stored",round(r.get("frac_of_stored_bytes",0),3))
except Exception as e:
    print("FAILED $w $dt $lib", e); print(open("$OUT/b.err").read()[-400:])
PY
done; done; done; done 2>&1 | tee $OUT/ab_tpb.txt
exit 0
