#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_tiles2; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for w in "coloring_100k f64 2000" "coloring_100k f32 2000" "coloring_10k f64 2000" "coloring_10k f32 2000" "ising_1024 f32 300" "coloring_1m_deg6 f32 200"; do
  set -- $w
  for kb in 0 128 256 512; do
    MAXSUM_TILE_KB=$kb timeout 200 python bench.py --no-cpu-baseline --configs main --workload $1 --dtype $2 --steps $3 --warmup $(($3/10)) 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(json.dumps({'workload':'$1','dtype':'$2','tile_kb':$kb,'us_per_cycle':round(d['ms_per_step']*1000,2),'frac':round(d['roofline']['frac'],4)}))" | tee -a $OUT/tiles.jsonl
  done
done
