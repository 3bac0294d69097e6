#!/bin/bash
# One A/B run on the GPU box, parametrised (replaces the one-off r03_exp*.sh scripts of round 3):
#
#   bash tools/ab.sh NAME "VARIANT;VARIANT;..." [--tests "tests/a.py tests/b.py"] -- COMMAND ...
#
# runs COMMAND once per VARIANT, a VARIANT being space-separated VAR=value assignments ("-" = the defaults), e.g.
#   bash tools/ab.sh rows "-;DFX_GEMM_NO_ROWS=1" -- env FRAMES=4 LIB=0 python tools/bench_gemm.py
#   bash tools/ab.sh tile "DFX_GEMM_TILE=0;DFX_GEMM_TILE=1;DFX_GEMM_TILE=7" -- python tools/rank_step.py
# and writes gpurun_out/$ROUND/NAME_<i>.txt (+ NAME_variants.txt).  With --tests the named GPU tests must pass first.
# A variant may select another build of the library with DFX_LIBRARY=/path/to/libdfx_variant.so (dfx/_lib.py): build it to a
# scratch path (hipcc ... -DDFX_... -o /tmp/x.so) - the shipped dfx/libdfx.so is never overwritten.
set -o pipefail
ROUND=${ROUND:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/$ROUND; mkdir -p $O
NAME=$1; VARIANTS=$2; shift 2
TESTS=""
if [ "$1" = "--tests" ]; then TESTS=$2; shift 2; fi
[ "$1" = "--" ] && shift
cd $R
if [ -n "$TESTS" ]; then
  python -m pytest $TESTS -x -q -m gpu > $O/${NAME}_tests.log 2>&1 || { tail -30 $O/${NAME}_tests.log; exit 1; }
  tail -2 $O/${NAME}_tests.log
fi
IFS=';' read -ra VS <<< "$VARIANTS"
: > $O/${NAME}_variants.txt
i=0
for v in "${VS[@]}"; do
  echo "$i: $v" >> $O/${NAME}_variants.txt
  if [ "$v" = "-" ]; then "$@" > $O/${NAME}_$i.txt 2>&1; else env $v "$@" > $O/${NAME}_$i.txt 2>&1; fi
  echo "== variant $i ($v): rc $?"; tail -${TAIL:-6} $O/${NAME}_$i.txt
  i=$((i+1))
done
