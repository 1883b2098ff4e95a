#!/bin/bash
# One GPU call, most important evidence first; every step skips itself once the call's
# time budget is used up (GPU minutes are rationed).
# usage: gpurun --timeout 800 -- 'bash scripts/gpu_final.sh TAG [BUDGET_SECONDS]'
TAG=${1:-final}
BUDGET=${2:-720}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
T0=$(date +%s)
el() { echo $(( $(date +%s) - T0 )); }
lap() { echo "== [$(el)s] $1"; }
left() { [ $(el) -lt $BUDGET ]; }
bench_line() {  # args... -> one json object per line in bench_variants.jsonl
  (echo -n "{\"args\": \"$*\", \"out\": "; timeout 300 python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1; echo "}") | tee -a $OUT/bench_variants.jsonl | cut -c1-330
}
pmc() {  # workload counter extra-args...
  local w=$1 c=$2; shift 2
  local d=$OUT/pmc_${w}_${c}$(echo "$*" | tr -d ' -')
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $d -o pmc -- python $R/bench.py --no-cpu-baseline --workload $w --steps 24 --warmup 6 --graph-chunk 0 "$@" > $d.log 2>&1
  local f=$(find $d -name "*counter_collection*.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py "$f" | grep -v rocclr | sed "s/^/$w $* /"
  rm -rf $d
}

lap "pytest -m gpu"
timeout 600 python -m pytest tests -x -q -m gpu --durations=5 2>&1 | tail -14 | tee $OUT/pytest_gpu.txt
lap smoke
timeout 200 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3 | tee $OUT/smoke.txt
lap "bench default (with the cpu baseline)"
timeout 400 python bench.py 2>&1 | tail -1 | tee $OUT/bench_default.json
: > $OUT/bench_variants.jsonl
lap "A/B: factor order (256 = caller's order inside a class, default = sorted by first variable)"
bench_line --layout-flags 256
bench_line
bench_line --workload coloring_1m_deg6 --steps 200 --warmup 20
bench_line --workload coloring_1m_deg6 --steps 200 --warmup 20 --layout-flags 256
lap "rocprofv3 kernel trace (default bench)"
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -o trace -- python $R/bench.py --no-cpu-baseline --steps 500 --warmup 50 > $OUT/prof_default.log 2>&1 )
tail -1 $OUT/prof_default.log | cut -c1-300
find $OUT/prof_default -name "*kernel_stats*.csv" | head -1 | while read f; do head -6 "$f"; cp "$f" $OUT/kernel_stats_default.csv; done
rm -rf $OUT/prof_default
if left; then lap "pmc: coloring_100k (separate passes)"
  ( cd /tmp; for c in FETCH_SIZE WRITE_SIZE; do pmc coloring_100k $c; done ) | tee $OUT/pmc_coloring_100k.txt
fi
if left; then lap "shard cost: what one rank of the 8-GPU / 2-GPU weak-scaling run does per cycle"
  (timeout 300 python tools/shard_cost.py 8 f64 2>&1 | tail -1; timeout 200 python tools/shard_cost.py 2 f64 2>&1 | tail -1) | tee $OUT/shard_cost.jsonl
fi
if left; then lap "pmc: coloring_1m_deg6"
  ( cd /tmp; for c in FETCH_SIZE WRITE_SIZE; do pmc coloring_1m_deg6 $c; done ) | tee $OUT/pmc_coloring_1m_deg6.txt
fi
if left; then lap "other workloads"
  bench_line --dtype f32
  bench_line --workload coloring_10k
  bench_line --workload coloring_100k_hard
  bench_line --workload ising_1024 --steps 300 --warmup 30
  bench_line --workload coloring_1m_deg6 --steps 200 --warmup 20 --dtype f32
fi
if left; then lap "pmc: unsorted factor order, coloring_100k FETCH_SIZE"
  ( cd /tmp; pmc coloring_100k FETCH_SIZE --layout-flags 256 ) | tee $OUT/pmc_coloring_100k_unsorted.txt
fi
if left; then lap "pmc calibration (tools/microbench: kernels of known traffic, incl. dense / half-dense 32-B gathers)"
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp; timeout 200 rocprofv3 --pmc $c --output-format csv -d $OUT/cal_$c -o pmc -- $R/tools/microbench > $OUT/cal_$c.log 2>&1 )
    f=$(find $OUT/cal_$c -name "*counter_collection*.csv" | head -1)
    [ -n "$f" ] && python $R/scripts/pmc_summary.py "$f" | grep -v -E "k_empty|k_chase|rocclr" | tee $OUT/pmc_calibration_$c.txt
    rm -rf $OUT/cal_$c
  done
  grep -h gather32 $OUT/cal_FETCH_SIZE.log | tee $OUT/microbench_gather32.jsonl
fi
if left; then lap "more workloads"
  bench_line --workload ising_1024 --steps 300 --warmup 30 --dtype f32
  bench_line --workload meeting_50k --steps 40 --warmup 5
  bench_line --layout-flags 32
  bench_line --layout-flags 64
fi
if left; then lap "pmc: ising_1024"
  ( cd /tmp; for c in FETCH_SIZE WRITE_SIZE; do pmc ising_1024 $c; done ) | tee $OUT/pmc_ising_1024.txt
fi
find $OUT -name "*.rocpd" -delete 2>/dev/null
lap done
du -sh $OUT
