set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python -m pytest tests/test_gemm_gpu.py tests/test_models_gpu.py -x -q -m gpu > $O/t_pair.log 2>&1 || { tail -30 $O/t_pair.log; exit 1; }
tail -2 $O/t_pair.log
DFX_WINO_WIDE=1 python -m pytest tests/test_conv_gpu.py -x -q -m gpu > $O/t_conv_wide.log 2>&1 || { tail -30 $O/t_conv_wide.log; exit 1; }
tail -2 $O/t_conv_wide.log
for w in 0 1 0 1; do DFX_WINO_WIDE=$w FRAMES=32 LIB=0 python tools/bench_conv.py > $O/conv_wide${w}_$RANDOM.txt 2>&1; echo "wide $w done"; done
for p in 0 1 0 1; do DFX_PAIR_SHORTCUT=$p python tools/stage_times.py 32 > $O/stage_pair${p}_$RANDOM.txt 2>&1; echo "pair $p done"; done
cd /tmp && export TMPDIR=/tmp
CASES="[(4,4,False)]" rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f4 -o run -- python3 $R/tools/rank_step.py > $O/stats_f4.log 2>&1
rm -f $O/stats_f4/*kernel_trace.csv
ls $O/stats_f4
