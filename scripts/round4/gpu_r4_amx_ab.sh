#!/bin/bash
# amaxsum: the running order of the destinations (static / by actual queue length / the default mix), three runs each
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4_amx_ab; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for mode in default dynamic static default dynamic static default dynamic static; do
  if [ $mode = default ]; then unset MAXSUM_AMAXSUM_ORDER; else export MAXSUM_AMAXSUM_ORDER=$mode; fi
  timeout 300 python tools/amaxsum_bench.py --no-oracle 100000 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(json.dumps({'order':'$mode','messages_per_s':d['messages_per_s'],'seconds':d['seconds'],'first_run':d['seconds_first_run']}))" | tee -a $OUT/ab.jsonl
done
