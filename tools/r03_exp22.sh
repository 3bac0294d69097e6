set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
for v in 0 33 0 50; do
  if [ $v = 0 ]; then FRAMES=32 LIB=0 python tools/bench_gemm.py > $O/gemm_stag_${v}_$RANDOM.txt 2>&1; else DFX_GEMM_STAGGER=$v FRAMES=32 LIB=0 python tools/bench_gemm.py > $O/gemm_stag_$v.txt 2>&1; fi
done
ls $O/gemm_stag_*
