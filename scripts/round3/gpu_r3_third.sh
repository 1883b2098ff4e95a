#!/bin/bash
# round 3, third call: wide kernel standalone, streaming policy A/B, barrier microbench, locality
# ordering A/B, new reference-vs-HIP tests; the reference's thread agents at 10k variables run in
# the background on the host cores meanwhile
TAG=${1:-r3_third}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
( timeout 900 python tools/reference_cpu_baseline.py --mode threads --timeout 120 --agents 1 --n-vars 10000 --out $OUT/reference_10k_k1.jsonl > $OUT/reference_10k_k1.log 2>&1 ) &
P1=$!
( timeout 900 python tools/reference_cpu_baseline.py --mode threads --timeout 120 --agents 64 --n-vars 10000 --out $OUT/reference_10k_k64.jsonl > $OUT/reference_10k_k64.log 2>&1 ) &
P2=$!
echo "== tests"
( time timeout 900 python -m pytest tests/test_gpu_vs_reference.py tests/test_gpu_plugin.py tests/test_gpu_reference_e2e.py -x -q -m gpu ) 2>&1 | tail -6 | tee $OUT/pytest_ref.txt
echo "== meeting_50k: launches one after the other (kernel times of their own), then overlapped"
cd /tmp
MAXSUM_NARY_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python $R/bench.py --no-cpu-baseline --configs main --workload meeting_50k --steps 200 --warmup 20 > $OUT/prof_meeting.log 2>&1
f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_meeting50k_f64_serial.csv && cut -c1-200 $OUT/kernel_stats_meeting50k_f64_serial.csv | head -4; rm -rf $OUT/p
MAXSUM_NARY_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python $R/bench.py --no-cpu-baseline --configs main --workload meeting_50k --dtype f32 --steps 200 --warmup 20 > $OUT/prof_meeting32.log 2>&1
f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_meeting50k_f32_serial.csv && cut -c1-200 $OUT/kernel_stats_meeting50k_f32_serial.csv | head -4; rm -rf $OUT/p
cd $R
for dt in f64 f32; do for ov in 0 1; do echo -n "overlap=$ov $dt: "; MAXSUM_NARY_OVERLAP=$ov timeout 300 python bench.py --no-cpu-baseline --configs main --workload meeting_50k --dtype $dt --steps 300 --warmup 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1),'us')"; done; done | tee $OUT/meeting_overlap_ab.txt
echo "== streaming policy A/B"
for w in coloring_1m_deg6 ising_1024 coloring_100k; do for dt in f64 f32; do for st in 0 1; do echo -n "$w $dt streaming=$st: "; MAXSUM_STREAMING=$st timeout 300 python bench.py --no-cpu-baseline --configs main --workload $w --dtype $dt --steps 400 --warmup 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_us'],2),'us frac',round(d['roofline']['frac'],3))"; done; done; done | tee $OUT/streaming_ab.txt
echo "== barrier bench"
for nb in 236 512 1024; do timeout 120 tools/barrier_bench $nb; done | tee $OUT/barrier_bench.jsonl
echo "== locality ordering"
timeout 600 python tools/locality_ab.py coloring_1m_deg6 f32 f64 2>&1 | tail -3 | tee $OUT/locality_ab.jsonl
echo "== waiting for the reference runs"
wait $P1 $P2
cat $OUT/reference_10k_k1.jsonl $OUT/reference_10k_k64.jsonl | cut -c1-400
