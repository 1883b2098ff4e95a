#!/bin/bash
# tiled factor order: window size sweep ($MAXSUM_TILE_KB), colouring instances, f64 and f32
TAG=${1:-r4_tiles}; SIZES=${2:-"0 512 1024 2048 4096 8192 16384"}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for w in "coloring_1m_deg6 f64 200" "coloring_1m_deg6 f32 200" "coloring_100k f64 2000" "coloring_100k f32 2000"; do
  set -- $w
  for kb in $SIZES; do
    MAXSUM_TILE_KB=$kb timeout 200 python bench.py --no-cpu-baseline --configs main --workload $1 --dtype $2 --steps $3 --warmup $(($3/10)) 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(json.dumps({'workload':'$1','dtype':'$2','tile_kb':$kb,'us_per_cycle':round(d['ms_per_step']*1000,2),'frac':round(d['roofline']['frac'],4)}))" | tee -a $OUT/tiles.jsonl
  done
done
