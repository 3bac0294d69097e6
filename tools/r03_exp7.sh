set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
for p in 0 1 0 1; do DFX_LINEAR_LN=$p python tools/stage_times.py 32 2>&1 | grep -E "spatial"; done
