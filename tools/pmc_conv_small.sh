set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04/pmc_small; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/p1 -o run -- python3 $R/tools/pmc_conv_small.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --output-format csv -d $O/p2 -o run -- python3 $R/tools/pmc_conv_small.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -o run -- python3 $R/tools/pmc_conv_small.py > /dev/null 2>&1
python3 $R/tools/pmc_report.py $O/p1 $O/p2 $O/p3 > $R/gpurun_out/r04/pmc_conv_small.md 2>&1
rm -rf $O
cat $R/gpurun_out/r04/pmc_conv_small.md
