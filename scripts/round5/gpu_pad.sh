#!/bin/bash
# Round 5: how far the box kernel's lane grid may overhang the table (padded cells at most 150 % / 240 % of the table's).
TAG=${1:-r5_pad}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for dt in f64 f32; do for lib in libmaxsum_hip.so libmaxsum_hip_pad240.so; do
  MAXSUM_HIP_LIB=$R/pydcop_amd/csrc/$lib timeout 400 python bench.py --workload meeting_50k_hetero --dtype $dt --configs main --no-cpu-baseline --steps 100 --warmup 10 > $OUT/b.json 2> $OUT/b.err
  python - <<PY
import json
d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("meeting_50k_hetero $dt $lib", round(d["ms_per_step"]*1e3,1),"us  stored",round(r.get("frac_of_stored_bytes",0),3), d["config"]["factor_kernels"])
PY
done; done 2>&1 | tee $OUT/ab.txt
exit 0
