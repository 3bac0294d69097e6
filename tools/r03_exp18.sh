# forced-tile sweep at 32 frames under the lean epilogue
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
for t in d 0 1 2 7; do
  if [ $t = d ]; then FRAMES=32 LIB=0 python tools/bench_gemm.py > $O/gemm_F32e_tile_d.txt 2>&1
  else DFX_GEMM_TILE=$t FRAMES=32 LIB=0 python tools/bench_gemm.py > $O/gemm_F32e_tile_$t.txt 2>&1; fi
done; echo done
