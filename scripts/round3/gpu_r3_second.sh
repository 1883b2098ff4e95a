#!/bin/bash
# round 3, second call: the rewritten wide-variable kernel (parity at full size, timing, kernel
# trace of meeting_50k) and the cache-policy variants of the sweep
TAG=${1:-r3_second}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
echo "== parity (whole parity file + fuzz)"
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_compile_path.py -x -q -m gpu --durations=5 ) 2>&1 | tail -14 | tee $OUT/pytest_parity.txt
echo "== meeting_50k"
for dt in f64 f32; do timeout 300 python bench.py --no-cpu-baseline --configs main --workload meeting_50k --dtype $dt --steps 300 --warmup 30 2>&1 | tail -1 | tee $OUT/bench_meeting_$dt.json | cut -c1-700; done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python $R/bench.py --no-cpu-baseline --configs main --workload meeting_50k --steps 200 --warmup 20 > $OUT/prof_meeting.log 2>&1
f=$(find $OUT/p -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_meeting50k_f64.csv && cut -c1-220 $OUT/kernel_stats_meeting50k_f64.csv | head -8; rm -rf $OUT/p
cd $R
echo "== cache policy variants"
bash scripts/gpu_ab_lib.sh $TAG/nt "libmaxsum_hip.so libmaxsum_hip_nt1.so libmaxsum_hip_nt2.so libmaxsum_hip_nt3.so libmaxsum_hip_nt8.so libmaxsum_hip_nt11.so libmaxsum_hip_nt15.so" "--configs main --steps 2000 --warmup 200" "--configs main --workload coloring_1m_deg6 --steps 300 --warmup 30" "--configs main --workload ising_1024 --steps 500 --warmup 50" "--configs main --workload coloring_1m_deg6 --dtype f32 --steps 300 --warmup 30" 2>&1 | tee $OUT/nt_ab.txt
