#!/bin/bash
# Round 6: the random-instance sweep of tests/fuzz_common.py on the GPU beyond the seeds of the test suite: fuzz_maxsum on seeds
# FIRST..LAST-1 with the default domains (arity 4 over 17 values: the multi-pass workgroup kernel among the classes) and again
# with FUZZ_DOMS=big (domains to 33), the HIP engine against the oracle bit for bit.
TAG=${1:-r6_fuzz}; FIRST=${2:-100}; LAST=${3:-1100}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
cat > /tmp/gpu_fuzz.py <<PY
import os, sys, time
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
from oracle.maxsum_oracle import build as build_oracles
build_oracles()
from fuzz_common import fuzz_maxsum
first, last = int(sys.argv[1]), int(sys.argv[2])
bad, t0 = 0, time.time()
for s in range(first, last):
    try:
        fuzz_maxsum(s, None)
    except Exception as ex:
        bad += 1
        print("FAIL seed", s, repr(ex)[:300], flush=True)
print("FUZZ_DOMS=%s seeds %d..%d: %d failures in %.0f s" % (os.environ.get("FUZZ_DOMS", "default"), first, last - 1, bad, time.time() - t0), flush=True)
PY
( timeout 1500 python3 /tmp/gpu_fuzz.py $FIRST $LAST; FUZZ_DOMS=big timeout 1500 python3 /tmp/gpu_fuzz.py $FIRST $LAST ) 2>&1 | tail -20 | tee $OUT/fuzz.txt
exit 0
