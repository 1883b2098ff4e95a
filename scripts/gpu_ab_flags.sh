#!/bin/bash
# A/B of layout flags on the bench workloads.  usage: gpu_ab_flags.sh TAG "flagsA flagsB ..." [small]
# (PMC passes: gpu_pmc.sh, one counter per pass)
TAG=$1; FLAGS=$2; SMALL=$3
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
: > $OUT/ab.jsonl
for rep in 1 2; do
for w in "coloring_100k --steps 2000 --warmup 200" "coloring_100k --dtype f32 --steps 2000 --warmup 200" \
         "coloring_10k --steps 2000 --warmup 200" \
         "ising_1024 --steps 300 --warmup 30" "ising_1024 --dtype f32 --steps 300 --warmup 30" \
         "coloring_1m_deg6 --steps 200 --warmup 20" "coloring_1m_deg6 --dtype f32 --steps 200 --warmup 20"; do
  if [ -n "$SMALL" ]; then case "$w" in coloring_10*) ;; *) continue ;; esac; fi
  for f in $FLAGS; do
    (echo -n "{\"w\": \"$w\", \"flags\": $f, \"out\": "; timeout 600 python bench.py --no-cpu-baseline --configs main --workload $w --layout-flags $f 2>&1 | tail -1 | tr -d "\n"; echo "}") >> $OUT/ab.jsonl
  done
done
done
python - $OUT/ab.jsonl <<'PY'
import json, sys
for line in open(sys.argv[1]):
    try:
        d = json.loads(line)
        o = d["out"]
        print(f'{d["w"]:60s} flags {d["flags"]:5d}  {o["roofline"]["avg_launch_us"]:9.2f} us  frac {o["roofline"]["frac"]:.3f}')
    except Exception as e:
        print("??", line[:200])
PY
