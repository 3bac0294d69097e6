#!/bin/bash
# Round evidence, part A (bench line, rocprofv3 kernel stats, PMC passes): ROUND=r04 bash tools/profile_round_a.sh
# (part B = tools/profile_round_b.sh: microbenchmarks and stage tables; together they are tools/profile_round.sh split to fit one
# gpurun call each)
set -o pipefail
ROUND=${ROUND:-r04}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${ROUND}prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default_line.json 2> $O/bench_default.err; echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --no-cpu-baseline > $O/stats.log 2>&1; echo "stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_p0 -o run -- python3 $R/bench.py --no-cpu-baseline --pipeline 0 > $O/stats_p0.log 2>&1; echo "stats p0 done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --no-cpu-baseline --pipeline 0 --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/bench.py --no-cpu-baseline --pipeline 0 --steps 2 --warmup 1 > /dev/null 2>&1; echo "traffic done"
rm -f $O/stats/*kernel_trace.csv $O/stats_p0/*kernel_trace.csv
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 3 $O/pmc_traffic.json > $O/pmc_traffic_rows.md 2>&1
rm -rf $O/pmc_fetch $O/pmc_write
cat $O/pmc_traffic_rows.md; ls $O/stats $O/stats_p0
