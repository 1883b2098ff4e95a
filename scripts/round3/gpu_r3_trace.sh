#!/bin/bash
# do the kernel trace and the HIP-event time of ONE run agree?  per-dispatch durations of the timed launches
TAG=${1:-r3_trace}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o trace -- python $R/bench.py --no-cpu-baseline --configs main --steps 2000 --warmup 200 > $OUT/prof.log 2>&1
tail -1 $OUT/prof.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench line: ms_per_step', d['ms_per_step'], 'event avg_launch_us', d['roofline']['avg_launch_us'])"
f=$(find $OUT/p -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $OUT/trace_vs_events.txt
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k_sweep' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
dur=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows]
gap=[int(rows[i+1]['Start_Timestamp'])-int(rows[i]['End_Timestamp']) for i in range(len(rows)-1)]
n=len(dur); timed=dur[-2000:]; g=gap[-1999:]
span=(int(rows[-1]['End_Timestamp'])-int(rows[-2000]['Start_Timestamp']))/2000
import statistics as st
print(f"dispatches {n}; all: avg {sum(dur)/n:.1f} ns; timed 2000: avg {sum(timed)/2000:.1f} ns, median {st.median(timed)} ns")
print(f"gaps between the timed dispatches: avg {sum(g)/len(g):.1f} ns, median {st.median(g)} ns")
print(f"first start to last end of the timed dispatches / 2000: {span:.1f} ns per cycle")
PY
cp $(find $OUT/p -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv; rm -rf $OUT/p
