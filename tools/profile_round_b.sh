#!/bin/bash
# Round evidence, part B (PMC table of the MFMA kernels, microbenchmarks, stage tables): ROUND=r04 bash tools/profile_round_b.sh
set -o pipefail
ROUND=${ROUND:-r04}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${ROUND}prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/pmc1 -o run -- python3 $R/tools/pmc_kernels.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --output-format csv -d $O/pmc2 -o run -- python3 $R/tools/pmc_kernels.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc3 -o run -- python3 $R/tools/pmc_kernels.py > /dev/null 2>&1; echo "pmc done"
python3 $R/tools/pmc_report.py $O/pmc1 $O/pmc2 $O/pmc3 > $O/pmc_kernels_table.md 2>&1
rm -rf $O/pmc1 $O/pmc2 $O/pmc3
cd $R
FRAMES=32 python3 tools/bench_gemm.py > $O/bench_gemm_F32.txt 2>&1
FRAMES=32 python3 tools/bench_conv.py > $O/bench_conv_F32.txt 2>&1
FRAMES=8 LIB=0 python3 tools/bench_conv.py > $O/bench_conv_F8.txt 2>&1; echo "micro done"
python3 tools/gemm_in_step.py 32 > $O/gemm_in_step_F32.txt 2>&1
python3 tools/stage_times.py 32 > $O/stage_times_mb32.txt 2>&1
python3 tools/stage_times.py 4 > $O/stage_times_mb4.txt 2>&1
python3 tools/conv_error.py > $O/conv_error.txt 2>&1
python3 tools/level_time.py 2.5 > $O/level_time.txt 2>&1; echo "all done"
cat $O/pmc_kernels_table.md
