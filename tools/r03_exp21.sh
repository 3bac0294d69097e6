# transposed accumulators + direct stores (DFX_GEMM_TR=1) vs the LDS round trip: tests, microbenchmark
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
DFX_GEMM_TR=1 timeout -k 10 400 python -m pytest tests/test_gemm_gpu.py tests/test_conv_gpu.py -x -q -m gpu > $O/t_exp21.log 2>&1 || { tail -30 $O/t_exp21.log; exit 1; }
tail -1 $O/t_exp21.log
FRAMES=32 LIB=0 python tools/bench_gemm.py > $O/gemm_tr0.txt 2>&1
DFX_GEMM_TR=1 FRAMES=32 LIB=0 python tools/bench_gemm.py > $O/gemm_tr1.txt 2>&1
FRAMES=32 LIB=0 python tools/bench_gemm.py > $O/gemm_tr0b.txt 2>&1
paste -d'|' <(grep -E "M=|Ci=" $O/gemm_tr0.txt | cut -c1-40,70-125) <(grep -E "M=|Ci=" $O/gemm_tr1.txt | cut -c70-125) <(grep -E "M=|Ci=" $O/gemm_tr0b.txt | cut -c70-125)
