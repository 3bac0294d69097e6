set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python -m pytest tests/test_gemm_gpu.py tests/test_detector_golden.py -x -q -m gpu > $O/t_exp13.log 2>&1 || { tail -30 $O/t_exp13.log; exit 1; }
tail -2 $O/t_exp13.log
python tools/stage_times.py 32 > $O/stage_times_mb32_r4800.txt 2>&1; DFX_GEMM_ROWS_MAX=9600 python tools/stage_times.py 32 > $O/stage_times_mb32_r9600.txt 2>&1
python tools/stage_times.py 4 > $O/stage_times_mb4_narrow.txt 2>&1
for f in $O/stage_times_mb32_r4800.txt $O/stage_times_mb32_r9600.txt $O/stage_times_mb4_narrow.txt; do echo $f; sed -n 9,11p $f; done
