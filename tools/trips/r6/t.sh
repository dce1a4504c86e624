#!/bin/bash
# trip t: balanced causal pairs after the latency work (Q prefetch behind the last tile, deferred flag, early flag poll): stamps + a shape matrix for the gate
O=gpurun_out/r6t; mkdir -p $O
L=universal-metal-flash-attention_amd/lib/libMFAFFI.so
for d in 0 1; do echo "== cbal delta $d"; timeout 60 tools/lab_bin/fwd_lab_c1 16 1024 30 4 $d | grep -v "^ *[0-9]"; done 2>&1 | tee $O/stamps_cbal.txt
for d in 0 1 2; do cp $L tools/lab_bin/libMFAFFI_cb$d.so; done
for D in 64 128; do for S in 512 1024 2048 4096; do for items in 128 256 512 1024 2048; do
  BH=$(( items * 128 / S )); [ $BH -lt 1 ] && continue; [ $BH -gt 256 ] && continue
  H=8; [ $(( BH % 8 )) -ne 0 ] && H=$BH; B=$(( BH / H ))
  timeout 120 python3 tools/ab_inproc.py --shape $B,$H,$S,$D --causal --out fp32 --graph --inner 50 --rounds 8 "off=$L:cbal=2,no_w64=1" "d0=tools/lab_bin/libMFAFFI_cb0.so:cbal=1,cbal_delta=0,no_w64=1" "d1=tools/lab_bin/libMFAFFI_cb1.so:cbal=1,cbal_delta=1,no_w64=1" "d2=tools/lab_bin/libMFAFFI_cb2.so:cbal=1,cbal_delta=2,no_w64=1" 2>&1 | grep -E "shape|Error|error|assert" | python3 -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()); continue
    print(json.dumps({'shape': r['shape'], 'items': $items, 'kernel': r['off']['kernel'], 'off': r['off']['ms_median'], 'd0': r['d0']['ms_median'], 'd1': r['d1']['ms_median'], 'd2': r['d2']['ms_median']}))
" | tee -a $O/matrix.jsonl
done; done; done
