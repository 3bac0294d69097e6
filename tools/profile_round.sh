#!/bin/bash
# Round-2 evidence in one GPU call: bash tools/profile_round.sh   (writes under gpurun_out/r02/)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default_line.json 2> $O/bench_default.err; echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --no-cpu-baseline > $O/stats.log 2>&1; echo "stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_p0 -o run -- python3 $R/bench.py --no-cpu-baseline --pipeline 0 > $O/stats_p0.log 2>&1; echo "stats p0 done"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/pmc1 -o run -- python3 $R/tools/pmc_kernels.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --output-format csv -d $O/pmc2 -o run -- python3 $R/tools/pmc_kernels.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc3 -o run -- python3 $R/tools/pmc_kernels.py > /dev/null 2>&1; echo "pmc done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --no-cpu-baseline --pipeline 0 --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/bench.py --no-cpu-baseline --pipeline 0 --steps 2 --warmup 1 > /dev/null 2>&1; echo "traffic done"
FRAMES=32 python3 $R/tools/bench_gemm.py > $O/bench_gemm_F32.txt 2>&1
FRAMES=8 python3 $R/tools/bench_gemm.py > $O/bench_gemm_F8.txt 2>&1
FRAMES=32 python3 $R/tools/bench_conv.py > $O/bench_conv_F32.txt 2>&1
FRAMES=8 python3 $R/tools/bench_conv.py > $O/bench_conv_F8.txt 2>&1; echo "micro done"
CASES="[(32,32,False),(16,16,False),(8,8,False),(4,4,False)]" python3 $R/tools/rank_step.py > $O/rank_step.txt 2>&1
python3 $R/tools/stage_times.py 32 > $O/stage_times_mb32.txt 2>&1
python3 $R/tools/conv_error.py > $O/conv_error.txt 2>&1; echo "all done"
# the raw per-dispatch CSVs are large: keep the summaries
rm -f $O/stats/*kernel_trace.csv $O/stats_p0/*kernel_trace.csv
python3 $R/tools/pmc_report.py $O/pmc1 $O/pmc2 $O/pmc3 > $O/pmc_kernels_table.md 2>&1
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 3 > $O/pmc_traffic_table.md 2>&1
rm -rf $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc_fetch $O/pmc_write
ls -la $O
