set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
DFX_GEMM_TILE=7 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu > $O/t_gemm_t7.log 2>&1 || { tail -30 $O/t_gemm_t7.log; exit 1; }
tail -2 $O/t_gemm_t7.log
for t in x 7 x 7; do if [ $t = x ]; then FRAMES=32 LIB=0 python tools/bench_gemm.py > $O/gemm_tile_def_$RANDOM.txt 2>&1; else DFX_GEMM_TILE=7 FRAMES=32 LIB=0 python tools/bench_gemm.py > $O/gemm_tile_7_$RANDOM.txt 2>&1; fi; echo "tile $t done"; done
