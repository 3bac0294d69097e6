# rows kernel: tests, microbenchmark lines at 4 / 32 frames with / without it, rank steps
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python -m pytest tests/test_gemm_gpu.py tests/test_models_gpu.py tests/test_detector_golden.py -x -q -m gpu > $O/t_exp11.log 2>&1 || { tail -30 $O/t_exp11.log; exit 1; }
tail -2 $O/t_exp11.log
for f in 4 32; do
  FRAMES=$f LIB=0 python tools/bench_gemm.py 2>&1 | grep -E "dec 300|qproj" > $O/rows_F${f}_on.txt
  DFX_GEMM_NO_ROWS=1 FRAMES=$f LIB=0 python tools/bench_gemm.py 2>&1 | grep -E "dec 300|qproj" > $O/rows_F${f}_off.txt
  cat $O/rows_F${f}_on.txt $O/rows_F${f}_off.txt
done
python tools/stage_times.py 4 > $O/stage_times_mb4_rows.txt 2>&1; DFX_GEMM_NO_ROWS=1 python tools/stage_times.py 4 > $O/stage_times_mb4_norows.txt 2>&1
python tools/stage_times.py 32 > $O/stage_times_mb32_rows.txt 2>&1; DFX_GEMM_NO_ROWS=1 python tools/stage_times.py 32 > $O/stage_times_mb32_norows.txt 2>&1
for f in $O/stage_times_mb4_rows.txt $O/stage_times_mb4_norows.txt $O/stage_times_mb32_rows.txt $O/stage_times_mb32_norows.txt; do echo $f; sed -n 9,11p $f; done
