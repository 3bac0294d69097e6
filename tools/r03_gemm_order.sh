set -o pipefail
O=gpurun_out/r03; mkdir -p $O
python -m pytest tests/test_gemm_gpu.py -x -q -m gpu > $O/t_gemm_g0.log 2>&1 || { tail -20 $O/t_gemm_g0.log; exit 1; }
DFX_GEMM_GROUP=4 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu > $O/t_gemm_g4.log 2>&1 || { tail -20 $O/t_gemm_g4.log; exit 1; }
for g in 0 1 2 4 8 16 1000; do
  DFX_GEMM_GROUP=$g FRAMES=32 LIB=0 python tools/bench_gemm.py > $O/order_g$g.txt 2>&1
  echo "group $g done"
done
