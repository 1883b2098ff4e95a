#!/bin/bash
# per-dispatch durations of the amaxsum generation kernels (100k-variable colouring, 16 generations)
TAG=${1:-r4_amx}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/p -o trace -- python $R/tools/amaxsum_bench.py --no-oracle 100000 > $OUT/prof.log 2>&1
f=$(find $OUT/p -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $OUT/amx_dispatches.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
tot = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "rocprim" in k: k = "rocprim::" + k.split("wrapped_")[-1].split("<")[0][:28]
    tot[k] = tot.get(k, 0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("total kernel ms", round(sum(tot.values()), 2), " span ms", round((max(int(r["End_Timestamp"]) for r in rows) - t0) / 1e6, 2))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]: print(f"   {k[-60:]:60s} {v:8.2f} ms")
gen = -1
for r in rows:
    n = r["Kernel_Name"]
    short = n.split("(")[0].replace("void ", "")
    if "rocprim" in short: short = "rocprim::" + short.split("wrapped_")[-1].split("<")[0][:28]
    if "k_dest" in short: gen += 1
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d >= 100 or "k_process" in short:
        print(f"gen {gen:2d} t {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us  {short[-48:]:48s} grid {r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size','?'):>9s}  {d:9.1f} us")
PY
rm -rf $OUT/p
