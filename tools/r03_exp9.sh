# forced-tile sweep of the GEMM microbenchmark at 4 and 8 frames (the rank block of an 8 / 4 GPU run)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
for f in 4 8; do
  for t in d 0 1 2 5 7; do
    if [ $t = d ]; then FRAMES=$f LIB=0 python tools/bench_gemm.py > $O/gemm_F${f}_tile_d.txt 2>&1
    else DFX_GEMM_TILE=$t FRAMES=$f LIB=0 python tools/bench_gemm.py > $O/gemm_F${f}_tile_$t.txt 2>&1; fi
  done; echo "frames $f done"
done
