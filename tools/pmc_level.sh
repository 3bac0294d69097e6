# PMC passes over the MSDA level kernel: ROUND=r04 bash tools/pmc_level.sh -> gpurun_out/$ROUND/pmc_level.txt
set -o pipefail
ROUND=${ROUND:-r04}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$ROUND; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for F in 8 32; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pl_fetch$F -o run -- python3 $R/tools/pmc_probe.py --frames $F > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pl_write$F -o run -- python3 $R/tools/pmc_probe.py --frames $F > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pl_sq$F -o run -- python3 $R/tools/pmc_probe.py --frames $F > /dev/null 2>&1
  echo "== N = $F frames" >> $O/pmc_level.txt
  python3 $R/tools/pmc_parse.py $O/pl_fetch$F $O/pl_write$F $O/pl_sq$F >> $O/pmc_level.txt 2>&1
  rm -rf $O/pl_fetch$F $O/pl_write$F $O/pl_sq$F
done
cat $O/pmc_level.txt
