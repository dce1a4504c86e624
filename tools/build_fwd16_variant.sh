#!/bin/bash
# Build a variant of libMFAFFI.so whose 128-row forward kernels (fa_fwd_16.hip, fa_fwd_16_pv.hip) are compiled with extra flags:
#   tools/build_fwd16_variant.sh NAME -DUMFA_D64_FORMS [-D...]   ->  tools/lab_bin/libMFAFFI_NAME.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/universal-metal-flash-attention_amd/csrc
NAME=$1; shift
TMP=$(mktemp -d)
mkdir -p $ROOT/tools/lab_bin
for f in fa_fwd_16 fa_fwd_16_pv; do
  ( cd $CS && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -w "$@" -c $f.hip -o $TMP/$f.o ) &
done
wait
OBJS=""
for f in runtime runtime_train fa_fwd_exact fa_fwd_wide fa_fwd16_w64 fa_fwd16_w64_bias fa_bwd fa_bwd_wide fa_bwd_16 fa_quant fa_aux tuning; do OBJS="$OBJS $CS/build/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$CS/exports.map -o $ROOT/tools/lab_bin/libMFAFFI_$NAME.so $OBJS $TMP/fa_fwd_16.o $TMP/fa_fwd_16_pv.o
rm -rf $TMP
echo built tools/lab_bin/libMFAFFI_$NAME.so
