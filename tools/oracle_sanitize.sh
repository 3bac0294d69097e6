#!/bin/bash
# The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: GPU sanitizers are not available on the
# pool): builds oracle/msda_oracle.c with -fsanitize=address,undefined into a scratch directory and runs the CPU tests that drive the oracle
# (golden vectors of the reference, edge cases) against that library.      bash tools/oracle_sanitize.sh [out.txt]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d)
trap 'rm -rf "$D"' EXIT
gcc -O1 -g -fPIC -std=c99 -Wall -Wextra -fno-fast-math -ffp-contract=off -fopenmp -fno-omit-frame-pointer \
    -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared -o "$D/libdfx_oracle.so" "$R/oracle/msda_oracle.c" -lm
export DFX_ORACLE_LIBRARY="$D/libdfx_oracle.so"
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=4
cd "$R"
python -m pytest tests/test_oracle_golden.py tests/test_models_golden.py tests/test_stream.py -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -5 | tee "${1:-/dev/null}"
