#!/bin/bash
# Build a timing-only variant of libMFAFFI.so with some filler kinds of the w64 stream dropped:
#   tools/build_w64_variant.sh NAME "EXP,ADD" [extra hipcc flags]   ->  tools/lab_bin/libMFAFFI_NAME.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/universal-metal-flash-attention_amd/csrc
NAME=$1; ABL=$2; shift 2
TMP=$(mktemp -d)
mkdir -p $ROOT/tools/lab_bin
cp $CS/*.hip $CS/*.h $CS/*.inc $TMP/
mkdir -p $TMP/../../include 2>/dev/null || true
W64_ABL="$ABL" W64_OUT=$TMP/fa_fwd16_w64_body.inc W64_OUT_I8=$TMP/fa_fwd_w64_i8_body.inc W64_OUT_I8F8=$TMP/fa_fwd_w64_i8f8_body.inc W64_OUT_D64=$TMP/fa_fwd16_w64d64_body.inc W64_OUT_BIAS=$TMP/fa_fwd16_w64_bias_body.inc W64_OUT_BIAS_D64=$TMP/fa_fwd16_w64d64_bias_body.inc python3 $ROOT/tools/gen_w64_body.py > /dev/null
# the generator also rewrites the helper include in-tree; it is identical for every variant
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fno-slp-vectorize -w "$@" -I$CS -c $TMP/fa_fwd16_w64.hip -o $TMP/fa_fwd16_w64.o
OBJS=""
for f in runtime runtime_train fa_fwd_exact fa_fwd_wide fa_fwd_16 fa_fwd_16_pv fa_fwd16_w64_bias fa_bwd fa_bwd_wide fa_bwd_16 fa_quant fa_aux tuning; do OBJS="$OBJS $CS/build/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/lab_bin/libMFAFFI_$NAME.so $OBJS $TMP/fa_fwd16_w64.o
rm -rf $TMP
echo built tools/lab_bin/libMFAFFI_$NAME.so
