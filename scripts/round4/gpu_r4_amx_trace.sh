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
# the warm run (from the second k_start_emit): time the GPU ran at least one kernel, and the idle gaps between
starts = [i for i, r in enumerate(rows) if "k_start_emit" in r["Kernel_Name"]]
if len(starts) >= 2:
    warm = rows[starts[1]:]
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in warm)
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    busy += cur_e - cur_s
    span = max(b for _, b in iv) - iv[0][0]
    print(f"warm run: span {span/1e6:.2f} ms, GPU busy {busy/1e6:.2f} ms, idle {(span-busy)/1e6:.2f} ms, {len(warm)} dispatches")
gen = -1
for r in rows:
    n = r["Kernel_Name"]
    short = n.split("(")[0].replace("void ", "")
    if "rocprim" in short: short = "rocprim::" + short.split("wrapped_")[-1].split("<")[0][:28]
    if "k_compact" in short: gen += 1
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d >= 100 or "k_process" in short:
        print(f"gen {gen:2d} t {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us  {short[-48:]:48s} grid {r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size','?'):>9s}  {d:9.1f} us")
PY
rm -rf $OUT/p
