#!/bin/bash
# Runs the child script of tests/test_clip_shard_gpu.py::test_rccl_call_path_with_one_rank N times and keeps each run's stderr:
# bash tools/rccl_one_rank_loop.sh [N=5] [outdir=gpurun_out/rccl_loop]
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-5}; O=${2:-$R/gpurun_out/rccl_loop}
mkdir -p "$O"
export DFX_PKG="$R/depth-fusion-in-transformer-based-video-object-detection_amd" DFX_ROOT="$R" HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$R"
for i in $(seq 1 "$N"); do
    export DFX_PORT=$((29600 + RANDOM % 300))
    python -c "import sys; sys.path.insert(0, '$R'); from tests.test_clip_shard_gpu import _RCCL_ONE_RANK as s; exec(s)" > "$O/out_$i.txt" 2> "$O/err_$i.txt"
    echo "run $i rc=$? $(tail -c 60 "$O/out_$i.txt" | tr '\n' ' ')"
done
