#!/bin/bash
# The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on this pool): builds an
# instrumented libsdpa_ref.so in a scratch directory, swaps it in for the run of the oracle's golden tests, restores the ordinary build.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
gcc -O1 -g -fPIC -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -std=c11 -shared -o $TMP/libsdpa_ref.so $ROOT/oracle/sdpa_ref.c -lm
make -C $ROOT/oracle -s
cp $ROOT/oracle/libsdpa_ref.so $TMP/orig.so
trap 'cp $TMP/orig.so $ROOT/oracle/libsdpa_ref.so; touch $ROOT/oracle/libsdpa_ref.so; rm -rf $TMP' EXIT
cp $TMP/libsdpa_ref.so $ROOT/oracle/libsdpa_ref.so; touch $ROOT/oracle/libsdpa_ref.so
cd $ROOT
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" python -m pytest tests/test_oracle_golden.py -x -q
