#!/bin/bash
# A/B of layout flags on the bench workloads.  usage: gpu_ab_flags.sh TAG "flagsA flagsB ..." [pmc]
TAG=$1; FLAGS=$2; PMC=$3
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
: > $OUT/ab.jsonl
for rep in 1 2; do
for w in "coloring_100k --steps 2000 --warmup 200" "coloring_100k --dtype f32 --steps 2000 --warmup 200" \
         "coloring_10k --steps 2000 --warmup 200" \
         "ising_1024 --steps 300 --warmup 30" "ising_1024 --dtype f32 --steps 300 --warmup 30" \
         "coloring_1m_deg6 --steps 200 --warmup 20" "coloring_1m_deg6 --dtype f32 --steps 200 --warmup 20"; do
  for f in $FLAGS; do
    (echo -n "{\"w\": \"$w\", \"flags\": $f, \"out\": "; timeout 600 python bench.py --no-cpu-baseline --configs main --workload $w --layout-flags $f 2>&1 | tail -1; echo "}") >> $OUT/ab.jsonl
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
if [ -n "$PMC" ]; then
cd /tmp
for w in coloring_100k coloring_1m_deg6 ising_1024; do
for f in $FLAGS; do
  timeout 600 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $OUT/p_${w}_$f -o pmc -- python $R/bench.py --no-cpu-baseline --configs main --workload $w --steps 30 --warmup 10 --graph-chunk 0 --layout-flags $f > $OUT/pmc_${w}_$f.log 2>&1
  c=$(find $OUT/p_${w}_$f -name "*counter_collection*.csv" | head -1)
  [ -n "$c" ] && python $R/scripts/pmc_summary.py "$c" | grep -v rocclr | sed "s/^/$w flags=$f /" | tee $OUT/pmc_${w}_$f.txt
  rm -rf $OUT/p_${w}_$f
done
done
fi
