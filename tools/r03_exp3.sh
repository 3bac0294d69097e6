set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python tools/wino_stamp.py > $O/wino_stamp.txt 2>&1; tail -60 $O/wino_stamp.txt
for g in 0 8 0 8; do DFX_GEMM_GROUP=$g python tools/stage_times.py 32 > $O/stage_group${g}_$RANDOM.txt 2>&1; echo "group $g done"; done
cd /tmp && export TMPDIR=/tmp
WEAK=0 CASES="[(4,4,False)]" rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f4 -o run -- python3 $R/tools/rank_step.py > $O/stats_f4.log 2>&1
rm -f $O/stats_f4/*kernel_trace.csv
