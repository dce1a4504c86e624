#!/bin/bash
# Build a variant of libMFAFFI.so whose additive-mask kernels (fa_fwd16_w64_bias.hip) use a differently generated body:
#   W64_DMA_M=... W64_MA_DEPTH=... tools/build_bias_variant.sh NAME [extra hipcc flags]   ->  tools/lab_bin/libMFAFFI_NAME.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/universal-metal-flash-attention_amd/csrc
NAME=$1; shift
TMP=$(mktemp -d)
mkdir -p $ROOT/tools/lab_bin
cp $CS/*.hip $CS/*.h $CS/*.inc $TMP/
W64_OUT=$TMP/x1.inc W64_OUT_I8=$TMP/x2.inc W64_OUT_I8F8=$TMP/x3.inc W64_OUT_D64=$TMP/x4.inc W64_OUT_BIAS=$TMP/fa_fwd16_w64_bias_body.inc W64_OUT_BIAS_D64=$TMP/fa_fwd16_w64d64_bias_body.inc python3 $ROOT/tools/gen_w64_body.py > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -fno-slp-vectorize -w "$@" -I$CS -c $TMP/fa_fwd16_w64_bias.hip -o $TMP/bias.o
if /opt/rocm/lib/llvm/bin/llvm-objdump -d --offloading $TMP/bias.o 2>/dev/null | grep -q scratch_; then echo "WARNING: scratch in the variant"; fi
OBJS=""
for f in runtime runtime_train fa_fwd_exact fa_fwd_wide fa_fwd_16 fa_fwd_16_pv fa_fwd16_w64 fa_bwd fa_bwd_wide fa_bwd_16 fa_quant fa_aux tuning; do OBJS="$OBJS $CS/build/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$CS/exports.map -o $ROOT/tools/lab_bin/libMFAFFI_$NAME.so $OBJS $TMP/bias.o
rm -rf $TMP
echo built tools/lab_bin/libMFAFFI_$NAME.so
