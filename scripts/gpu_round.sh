#!/bin/bash
# One gpurun call: GPU parity tests, smoke, microbench, bench variants, rocprofv3 evidence.
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh [tag] [quick]'
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== rocm-smi" ; rocm-smi --showproductname 2>/dev/null | head -8
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
echo "== smoke"
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -5 | tee $OUT/smoke.txt
echo "== microbench"
timeout 300 ./tools/microbench 2>&1 | tee $OUT/microbench.jsonl
echo "== bench default"
timeout 600 python bench.py 2>&1 | tail -3 | tee $OUT/bench_default.json
echo "== bench variants (no cpu baseline)"
: > $OUT/bench_variants.jsonl
for v in "--dtype f32" "--layout-flags 32" "--layout-flags 64" "--graph-chunk 0" \
         "--workload coloring_10k" "--workload coloring_100k_hard" \
         "--workload ising_1024 --steps 300 --warmup 30" "--workload ising_1024 --steps 300 --warmup 30 --dtype f32" \
         "--workload ising_1024 --steps 300 --warmup 30 --layout-flags 32" "--workload ising_1024 --steps 300 --warmup 30 --layout-flags 64" \
         "--workload coloring_1m_deg6 --steps 200 --warmup 20" "--workload coloring_1m_deg6 --steps 200 --warmup 20 --dtype f32" \
         "--workload coloring_1m_deg6 --steps 200 --warmup 20 --layout-flags 32" "--workload coloring_1m_deg6 --steps 200 --warmup 20 --layout-flags 64" \
         "--workload meeting_50k --steps 20 --warmup 3" "--workload meeting_50k --steps 20 --warmup 3 --dtype f32"; do
  echo "-- $v"
  (echo -n "{\"args\": \"$v\", \"out\": "; timeout 600 python bench.py --no-cpu-baseline $v 2>&1 | tail -1; echo "}") | tee -a $OUT/bench_variants.jsonl
done
echo "== rocprofv3 kernel trace"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -o trace -- python $R/bench.py --no-cpu-baseline --steps 500 --warmup 50 > $OUT/prof_default.log 2>&1
tail -2 $OUT/prof_default.log
find $OUT/prof_default -name "*kernel_stats*.csv" | head -2 | while read f; do echo "## $f"; head -8 "$f"; cp "$f" $OUT/kernel_stats_default.csv; done
echo "== rocprofv3 pmc (separate passes)"
for w in coloring_100k ising_1024; do
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${w}_$c -o pmc -- python $R/bench.py --no-cpu-baseline --workload $w --steps 60 --warmup 10 --graph-chunk 0 > $OUT/pmc_${w}_$c.log 2>&1
  f=$(find $OUT/pmc_${w}_$c -name "*counter_collection*.csv" | head -1)
  echo "## $w $c -> $f"
  [ -n "$f" ] && (head -1 "$f"; grep k_sweep "$f" | tail -20) > $OUT/pmc_${w}_$c.csv && tail -2 $OUT/pmc_${w}_$c.csv | cut -c1-400
  rm -rf $OUT/pmc_${w}_$c
done
done
echo "== pmc calibration on kernels of known traffic (tools/microbench)"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/cal_$c -o pmc -- $R/tools/microbench > $OUT/cal_$c.log 2>&1
  f=$(find $OUT/cal_$c -name "*counter_collection*.csv" | head -1)
  [ -n "$f" ] && python - "$f" $c <<'PY' | tee $OUT/pmc_calibration_$c.txt
import csv, sys, collections
acc = collections.OrderedDict()
for row in csv.DictReader(open(sys.argv[1])):
    k = (row["Kernel_Name"][:40], row["Grid_Size"])
    acc.setdefault(k, []).append(float(row["Counter_Value"]))
for (k, g), v in acc.items():
    print(f"{sys.argv[2]} {k:40s} grid {g:>9s} dispatches {len(v):5d} mean_KiB {sum(v)/len(v):12.1f} last_KiB {v[-1]:12.1f}")
PY
  rm -rf $OUT/cal_$c
done
find $OUT -name "*.rocpd" -size +8M -delete 2>/dev/null
rm -rf $OUT/prof_default
du -sh $OUT
