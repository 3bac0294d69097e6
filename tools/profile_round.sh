#!/bin/bash
# Round evidence in one GPU call: ROUND=r04 bash tools/profile_round.sh   (writes under gpurun_out/$ROUND/)
set -o pipefail
ROUND=${ROUND:-r04}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${ROUND}prof; mkdir -p $O
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
python3 $R/tools/proto_bf16x3.py > $O/proto_bf16x3.txt 2>&1
python3 $R/tools/wino_stamp.py > $O/wino_stamp.txt 2>&1
python3 $R/tools/memcpy_origins.py 4 > $O/memcpy_origins_f4.txt 2>&1
python3 $R/tools/stage_times.py 32 > $O/stage_times_mb32.txt 2>&1
python3 $R/tools/conv_error.py > $O/conv_error.txt 2>&1
python3 $R/tools/level_time.py 2.5 > $O/level_time.txt 2>&1
python3 $R/tools/graph_step.py > $O/graph_step.txt 2>&1
python3 $R/tools/launch_bound.py > $O/launch_bound.txt 2>&1; echo "all done"
# the raw per-dispatch CSVs are large: keep the summaries
rm -f $O/stats/*kernel_trace.csv $O/stats_p0/*kernel_trace.csv
cat > $O/pmc_kernels_table.md <<'HDR'
# PMC counters of the hand-written MFMA kernels, 32 frames (round 3)

Three `rocprofv3 --kernel-trace --pmc ... --output-format csv` passes over `tools/pmc_kernels.py` (one counter set per pass, as
`MI355X_MICROARCH.md` prescribes; no `--stats`, no tracing domains), tabulated by `tools/pmc_report.py`:

    pass 1: SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
    pass 2: SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU
    pass 3: GRBM_GUI_ACTIVE   (clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time)

Rows in launch order: own kernel, then (for the GEMMs) the library kernel torch picks for the same shape - conv1x1 1024->2048,
512->2048, 1024->256 at 50x84, 64->256 at 200x334; Linear 134400x256x256, x1024x256, x256x1024, 9600x32768x256; then Winograd
64->64 at 200x334, 256->256 at 50x84, 512->512 dilation 2 (main + quarter-size launches), implicit GEMM 128->128 3x3/2 and the 7x7/2
stem.  "MFMA busy / CU busy" = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); SQ_INSTS_VALU counts the MFMAs too
(1.0 of the "VALU / MFMA" column is the MFMA itself; the library kernels use the 16x16x4 MFMA: twice the instructions per flop).
Kernel times are under the counters (5-15 % slower than unprofiled).

HDR
python3 $R/tools/pmc_report.py $O/pmc1 $O/pmc2 $O/pmc3 >> $O/pmc_kernels_table.md 2>&1
cat > $O/pmc_traffic_table.md <<'HDR'
# Fabric-side traffic of the bench's kernel families (round 3)

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python3 bench.py --no-cpu-baseline --pipeline 0 --steps 2 --warmup 1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -- python3 bench.py --no-cpu-baseline --pipeline 0 --steps 2 --warmup 1
    python tools/pmc_traffic.py FETCH_DIR WRITE_DIR 3

Both counters in KiB at the memory side of L2 (HBM and Infinity Cache hits alike); FETCH_SIZE doubled: gfx950 reports half of
the bytes of wide coalesced reads (`MI355X_MICROARCH.md`, calibrated in `r01_pmc_msda_level_N8.md`).  Per 32-frame step
(3 steps profiled: 1 warm-up + 2 timed).

HDR
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 3 >> $O/pmc_traffic_table.md 2>&1
rm -rf $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc_fetch $O/pmc_write
ls -la $O
