#!/bin/bash
# GPU_MAX_HW_QUEUES (the HIP runtime's hardware queues per process, default 4) against the number of pipeline lanes: every
# configuration in a process of its own (streams of earlier runners would occupy queues).   bash tools/hw_queues_ab.sh > out.txt
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
one() {   # queues lanes frames
    GPU_MAX_HW_QUEUES=$1 WEAK=0 PIPE=1 GRAPH=1 LANES=$2 RUNNERS=1 PIPE_FRAMES="($3,)" CASES="[]" python tools/rank_step.py 2>&1 | grep pipelined | sed "s/^/queues $1: /"
}
bench1() {  # queues lanes
    GPU_MAX_HW_QUEUES=$1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-literal --lanes $2 2>/dev/null |
        python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $1: bench N=1, $2 eager lanes:', d['value'], 'frames/s', d['ms_per_step'], 'ms')"
}
for rep in 1 2; do
    bench1 4 2; bench1 8 2
    for f in 4 8 16; do one 4 3 $f; one 8 3 $f; one 8 4 $f; done
done
