set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
for t in d 0 1 2 5; do
  if [ $t = d ]; then python tools/bench_gemm_queries.py > $O/q_tile_d.txt 2>&1; else DFX_GEMM_TILE=$t python tools/bench_gemm_queries.py > $O/q_tile_$t.txt 2>&1; fi
done
DFX_GEMM_ROWS_MAX=9600 python tools/bench_gemm_queries.py > $O/q_tile_r.txt 2>&1
echo done
