#!/bin/bash
# Mid-round GPU call: parity tests, smoke, default bench, the HBM-sized variants,
# kernel-trace stats, PMC calibration on kernels of known traffic, PMC of the sweep.
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_mid.sh TAG'
TAG=${1:-mid}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
T0=$(date +%s)
lap() { echo "== [$(( $(date +%s) - T0 ))s] $1"; }
lap "pytest -m gpu"
timeout 900 python -m pytest tests -x -q -m gpu --durations=5 2>&1 | tail -14 | tee $OUT/pytest_gpu.txt
lap smoke
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3 | tee $OUT/smoke.txt
lap microbench
timeout 200 ./tools/microbench 2>&1 | tee $OUT/microbench.jsonl
lap "bench default"
timeout 600 python bench.py 2>&1 | tail -1 | tee $OUT/bench_default.json
lap "bench variants"
: > $OUT/bench_variants.jsonl
while IFS= read -r v; do
  echo "-- $v"
  (echo -n "{\"args\": \"$v\", \"out\": "; timeout 600 python bench.py --no-cpu-baseline $v 2>&1 | tail -1; echo "}") | tee -a $OUT/bench_variants.jsonl | cut -c1-400
done <<'EOF'
--dtype f32
--layout-flags 32
--layout-flags 64
--graph-chunk 0
--workload coloring_10k
--workload coloring_100k_hard
--workload ising_1024 --steps 300 --warmup 30
--workload ising_1024 --steps 300 --warmup 30 --dtype f32
--workload coloring_1m_deg6 --steps 200 --warmup 20
--workload coloring_1m_deg6 --steps 200 --warmup 20 --dtype f32
--workload meeting_50k --steps 80 --warmup 10
--workload meeting_50k --steps 80 --warmup 10 --dtype f32
EOF
lap "tools: timeline, boundary cost, shard cost"
timeout 300 python tools/timeline.py coloring_100k f64 2>&1 | grep -v "active blocks" | tail -8 | tee $OUT/timeline.txt
(timeout 300 python tools/boundary_cost.py coloring_100k 100 | tail -1; timeout 300 python tools/boundary_cost.py coloring_100k 2000 | tail -1) | tee $OUT/boundary_cost.jsonl
(timeout 600 python tools/shard_cost.py 2 f64 | tail -1; timeout 600 python tools/shard_cost.py 8 f64 | tail -1) | tee $OUT/shard_cost.jsonl
lap "rocprofv3 kernel trace"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -o trace -- python $R/bench.py --no-cpu-baseline --steps 500 --warmup 50 > $OUT/prof_default.log 2>&1
tail -1 $OUT/prof_default.log | cut -c1-300
find $OUT/prof_default -name "*kernel_stats*.csv" | head -1 | while read f; do head -6 "$f"; cp "$f" $OUT/kernel_stats_default.csv; done
rm -rf $OUT/prof_default
lap "pmc calibration (tools/microbench: kernels of known traffic)"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/cal_$c -o pmc -- $R/tools/microbench > $OUT/cal_$c.log 2>&1
  f=$(find $OUT/cal_$c -name "*counter_collection*.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py "$f" | grep -v -E "k_empty|k_chase" | tee $OUT/pmc_calibration_$c.txt
  rm -rf $OUT/cal_$c
done
lap "pmc sweep kernels"
for w in coloring_100k coloring_1m_deg6 ising_1024 meeting_50k; do
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${w}_$c -o pmc -- python $R/bench.py --no-cpu-baseline --workload $w --steps 24 --warmup 6 --graph-chunk 0 > $OUT/pmc_${w}_$c.log 2>&1
  f=$(find $OUT/pmc_${w}_$c -name "*counter_collection*.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py "$f" | grep -v rocclr | tee $OUT/pmc_${w}_$c.txt
  rm -rf $OUT/pmc_${w}_$c
done
done
find $OUT -name "*.rocpd" -delete 2>/dev/null
lap done
du -sh $OUT
