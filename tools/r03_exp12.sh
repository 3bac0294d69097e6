# rows kernel + attention wave groups: tests incl. the sharding suite, stage times and rank steps at 4 / 8 frames
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python -m pytest tests/test_gemm_gpu.py tests/test_models_gpu.py tests/test_detector_golden.py tests/test_clip_shard_gpu.py tests/test_stream.py -x -q -m gpu > $O/t_exp12.log 2>&1 || { tail -30 $O/t_exp12.log; exit 1; }
tail -2 $O/t_exp12.log
python tools/stage_times.py 4 > $O/stage_times_mb4_g.txt 2>&1; DFX_MHA_GROUPS=1 python tools/stage_times.py 4 > $O/stage_times_mb4_g1.txt 2>&1
python tools/stage_times.py 8 > $O/stage_times_mb8_g.txt 2>&1; DFX_MHA_GROUPS=1 python tools/stage_times.py 8 > $O/stage_times_mb8_g1.txt 2>&1
for f in $O/stage_times_mb4_g.txt $O/stage_times_mb4_g1.txt $O/stage_times_mb8_g.txt $O/stage_times_mb8_g1.txt; do echo $f; sed -n 9,11p $f; done
WEAK=0 PIPE=1 CASES="[(16,16,False),(8,8,False),(4,4,False)]" python tools/rank_step.py > $O/rank_step_rows_groups.txt 2>&1
tail -6 $O/rank_step_rows_groups.txt
