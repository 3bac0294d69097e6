# lean epilogue: GEMM / conv / model tests, then the microbenchmark old vs new epilogue in one box
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python -m pytest tests/test_gemm_gpu.py tests/test_conv_gpu.py tests/test_models_gpu.py tests/test_detector_golden.py -x -q -m gpu > $O/t_exp15.log 2>&1 || { tail -30 $O/t_exp15.log; exit 1; }
tail -2 $O/t_exp15.log
FRAMES=32 LIB=0 python tools/bench_gemm.py > $O/gemm_epi_new.txt 2>&1
DFX_GEMM_OLD_EPILOGUE=1 FRAMES=32 LIB=0 python tools/bench_gemm.py > $O/gemm_epi_old.txt 2>&1
FRAMES=32 LIB=0 python tools/bench_gemm.py > $O/gemm_epi_new2.txt 2>&1
paste -d'|' <(grep -E "M=|Ci=" $O/gemm_epi_old.txt | cut -c1-40,95-125) <(grep -E "M=|Ci=" $O/gemm_epi_new.txt | cut -c95-125) <(grep -E "M=|Ci=" $O/gemm_epi_new2.txt | cut -c95-125)
