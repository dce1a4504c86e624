#!/bin/bash
# Lab: libMFAFFI variant with ONE source file recompiled with extra flags:
#   tools/build_obj_variant.sh NAME fa_quant "-DUMFA_LAB_QMUL"   ->  tools/lab_bin/libMFAFFI_NAME.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/universal-metal-flash-attention_amd/csrc
NAME=$1; SRC=$2; shift 2
TMP=$(mktemp -d)
mkdir -p $ROOT/tools/lab_bin
EXTRA=""
[ "$SRC" = fa_fwd16_w64 ] && EXTRA="-fno-slp-vectorize"
[ "$SRC" = fa_bwd_16 ] && EXTRA="-fno-slp-vectorize -mllvm -pragma-unroll-threshold=262144"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -w $EXTRA $@ -I$CS -c $CS/$SRC.hip -o $TMP/$SRC.o
OBJS=""
for f in runtime runtime_train fa_fwd_exact fa_fwd_16 fa_fwd_16_pv fa_fwd16_w64 fa_bwd fa_bwd_16 fa_quant fa_aux tuning; do
  if [ $f = $SRC ]; then OBJS="$OBJS $TMP/$f.o"; else OBJS="$OBJS $CS/build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/lab_bin/libMFAFFI_$NAME.so $OBJS
rm -rf $TMP
echo built tools/lab_bin/libMFAFFI_$NAME.so
