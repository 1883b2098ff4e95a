#!/bin/bash
# For a box that has BOTH an MI355X and a pyDCOP checkout (neither the build container nor the
# gpurun box has both): the unmodified `pydcop solve` CLI with `--algo maxsum_gpu` on the real
# HIP library, next to the reference's own maxsum.
# usage: PYDCOP=/path/to/pyDcop scripts/plugin_on_gpu.sh [instance.yaml]
R=$(cd "$(dirname "$0")/.." && pwd)
PYDCOP=${PYDCOP:-/root/reference}
INST=${1:-$PYDCOP/tests/instances/graph_coloring1.yaml}
export PYTHONPATH=$R:$PYDCOP:$PYTHONPATH
python -c 'import __graft_entry__ as g; g.build()' || exit 1
echo "== maxsum_gpu (MI355X)"
python -m pydcop_amd.plugin -t 20 solve --algo maxsum_gpu -p stop_cycle:30 -p noise:0 -d adhoc "$INST"
echo "== maxsum (reference, thread agents)"
python -m pydcop_amd.plugin -t 5 solve --algo maxsum -p noise:0 -d adhoc "$INST"
