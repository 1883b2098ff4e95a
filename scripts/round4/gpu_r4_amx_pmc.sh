#!/bin/bash
# amaxsum chain kernels: VALU instructions, busy cycles, waves (one counter per pass) -- are they issue-bound or memory-bound?
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_amx_pmc; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for c in SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU; do
  timeout 200 rocprofv3 --pmc $c --output-format csv -d $OUT/p -o pmc -- python $R/tools/amaxsum_bench.py --no-oracle 100000 > $OUT/log_$c.txt 2>&1
  f=$(find $OUT/p -name "*counter_collection*.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" $c <<'PY' | tee -a $OUT/amx_pmc.txt
import csv, sys, collections
acc = collections.OrderedDict()
for row in csv.DictReader(open(sys.argv[1])):
    if "k_process" not in row["Kernel_Name"]: continue
    k = row["Kernel_Name"].split("(")[0][-40:]
    acc.setdefault(k, []).append(float(row["Counter_Value"]))
for k, v in acc.items():
    v2 = sorted(v)[-6:]
    print(f"{sys.argv[2]:22s} {k:42s} n {len(v):3d} sum {sum(v):16.0f} six largest dispatches {[int(x) for x in v2]}")
PY
  else echo "FAILED $c"; tail -3 $OUT/log_$c.txt; fi
  rm -rf $OUT/p
done
