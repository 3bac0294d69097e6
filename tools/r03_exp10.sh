# after the 4/8-frame tile rule + shared top-k: tests, default-tile microbenchmark at 4 / 8 / 16 / 32 frames, rank steps
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python -m pytest tests/test_gemm_gpu.py tests/test_detector_golden.py tests/test_stream.py tests/test_clip_shard_gpu.py tests/test_models_gpu.py -x -q -m gpu > $O/t_exp10.log 2>&1 || { tail -30 $O/t_exp10.log; exit 1; }
tail -2 $O/t_exp10.log
for f in 4 8 16 32; do FRAMES=$f LIB=0 python tools/bench_gemm.py > $O/gemm_F${f}_after.txt 2>&1; done; echo "micro done"
CASES="[(32,32,False),(16,16,False),(8,8,False),(4,4,False)]" python tools/rank_step.py > $O/rank_step_after.txt 2>&1
WEAK=0 PIPE=1 CASES="[(4,4,False),(8,8,False),(16,16,False)]" python tools/rank_step.py > $O/rank_step_pipe_after.txt 2>&1
tail -5 $O/rank_step_after.txt; tail -4 $O/rank_step_pipe_after.txt
