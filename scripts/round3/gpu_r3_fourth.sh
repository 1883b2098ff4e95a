#!/bin/bash
# wide kernel again (phases), tiles-per-workgroup variants, parity of the new wide kernel
TAG=${1:-r3_fourth}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
bash scripts/gpu_wide_phases.sh $TAG/phases "libmaxsum_hip.so libmaxsum_hip_ws1.so libmaxsum_hip_ws8.so libmaxsum_hip_ws31.so"
echo "== parity of the wide classes"
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "meeting or wide or nary or corner or mixed or hub" ) 2>&1 | tail -5 | tee $OUT/pytest_wide.txt
echo "== tiles per workgroup"
bash scripts/gpu_ab_lib.sh $TAG/tiles "libmaxsum_hip.so libmaxsum_hip_tiles2.so libmaxsum_hip_tiles3.so libmaxsum_hip_tiles4.so" "--configs main --steps 3000 --warmup 300" "--configs main --dtype f32 --steps 3000 --warmup 300" "--configs main --workload coloring_1m_deg6 --steps 300 --warmup 30" "--configs main --workload ising_1024 --steps 500 --warmup 50" "--configs main --workload coloring_10k --steps 4000 --warmup 400" 2>&1 | tee $OUT/tiles_ab.txt
for dt in f64 f32; do echo -n "meeting $dt: "; timeout 300 python bench.py --no-cpu-baseline --configs main --workload meeting_50k --dtype $dt --steps 300 --warmup 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1),'us', d['roofline']['bytes_basis'], round(d['roofline']['frac'],3))"; done | tee $OUT/meeting.txt
