# end-of-round verification + the evidence that changed after profile_round.sh:  ROUND=r04 bash tools/final_round.sh
set -o pipefail
ROUND=${ROUND:-r04}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${ROUND}final; mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu > $O/t_full.log 2>&1; tail -2 $O/t_full.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default_line.json 2> $O/bench_default.err; echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_p0 -o run -- python3 $R/bench.py --no-cpu-baseline --pipeline 0 > $O/stats_p0.log 2>&1; echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --no-cpu-baseline --pipeline 0 --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/bench.py --no-cpu-baseline --pipeline 0 --steps 2 --warmup 1 > /dev/null 2>&1; echo "traffic done"
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 3 > $O/pmc_traffic_rows.md 2>&1
rm -rf $O/pmc_fetch $O/pmc_write; rm -f $O/stats/*kernel_trace.csv $O/stats_p0/*kernel_trace.csv
cd $R
python tools/gemm_in_step.py 32 > $O/gemm_in_step_F32.txt 2>&1
FRAMES=32 python tools/bench_conv.py > $O/bench_conv_F32.txt 2>&1
FRAMES=32 python tools/bench_gemm.py > $O/bench_gemm_F32.txt 2>&1
WEAK=0 PIPE=1 CASES="[(32,32,False),(16,16,False),(8,8,False),(4,4,False)]" python tools/rank_step.py > $O/rank_step.txt 2>&1
python tools/stage_times.py 32 > $O/stage_times_mb32.txt 2>&1
tail -8 $O/rank_step.txt; cat $O/pmc_traffic_rows.md
