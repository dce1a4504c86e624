#!/bin/bash
# timing-only ablations of the fp8 tile stream (results wrong by construction)
set -e
cd "$(dirname "$0")/.."
tools/build_w64_variant.sh f8_noexp "EXP"
tools/build_w64_variant.sh f8_nocvt "CVT8"
tools/build_w64_variant.sh f8_nofma "FMA"
tools/build_w64_variant.sh f8_nomax "MAX,DEC"
tools/build_w64_variant.sh f8_nosm "EXP,CVT8,FMA,MAX,DEC"
tools/build_w64_variant.sh f8_nodma "DMAK,DMAV,UPDK,UPDV" -DW64_ABL_NODMA
tools/build_w64_variant.sh f8_nobar "" -DW64_ABL_NOBAR
tools/build_w64_variant.sh f8_noreads "VREAD8,KREAD,KPRE"
tools/build_w64_variant.sh f8_mfmaonly "EXP,CVT8,FMA,MAX,DEC,DMAK,DMAV,UPDK,UPDV,VREAD8,KREAD" -DW64_ABL_NOBAR
