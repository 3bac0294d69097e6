set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python -m pytest tests/test_gemm_gpu.py tests/test_models_gpu.py tests/test_detector_golden.py tests/test_stream.py tests/test_configs_gpu.py -x -q -m gpu > $O/t_ln.log 2>&1 || { tail -40 $O/t_ln.log; exit 1; }
tail -2 $O/t_ln.log
for p in 0 1 0 1; do DFX_LINEAR_LN=$p python tools/stage_times.py 32 > $O/stage_ln${p}_$RANDOM.txt 2>&1; echo "ln $p"; grep -E "spatial|frame stage|temporal" $O/stage_ln${p}_*.txt | tail -3; done
for p in 0 1; do DFX_LINEAR_LN=$p WEAK=0 CASES="[(4,4,False),(32,32,False)]" python tools/rank_step.py 2>&1 | tail -2; done
