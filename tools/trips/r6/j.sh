#!/bin/bash
# trip j: same-box A/B of this round's library against round 5's (tools/lab_bin/libMFAFFI_r5.so) on kernels whose TEXT changed only incidentally
# (segment-scope lane ids, mk_steps order): headline, causal, head_dim 64, bool masks, int8 -- nothing here may have moved
O=gpurun_out/r6j; mkdir -p $O
L=universal-metal-flash-attention_amd/lib/libMFAFFI.so; R=tools/lab_bin/libMFAFFI_r5.so
python3 tools/ab_inproc.py --shape 1,24,4096,128 --out fp32 --graph --rounds 12 r5=$R r6=$L 2>&1 | grep shape | tee -a $O/ab_r5_vs_r6.jsonl
python3 tools/ab_inproc.py --shape 4,16,4096,128 --causal --graph --rounds 8 r5=$R r6=$L 2>&1 | grep shape | tee -a $O/ab_r5_vs_r6.jsonl
python3 tools/ab_inproc.py --shape 2,16,4096,64 --graph --rounds 8 r5=$R r6=$L 2>&1 | grep shape | tee -a $O/ab_r5_vs_r6.jsonl
python3 tools/ab_inproc.py --shape 4,16,1024,64 --causal --graph --inner 100 --rounds 8 r5=$R r6=$L 2>&1 | grep shape | tee -a $O/ab_r5_vs_r6.jsonl
python3 tools/ab_inproc.py --shape 1,24,4096,128 --out fp32 --graph --mask blockdiag --rounds 8 r5=$R r6=$L 2>&1 | grep shape | tee -a $O/ab_r5_vs_r6.jsonl
python3 tools/ab_inproc.py --shape 1,24,4096,128 --out fp32 --graph --mask padding --rounds 8 r5=$R r6=$L 2>&1 | grep shape | tee -a $O/ab_r5_vs_r6.jsonl
python3 tools/ab_inproc.py --shape 1,24,4096,128 --out fp32 --graph --mask bias --rounds 8 r5=$R r6=$L 2>&1 | grep shape | tee -a $O/ab_r5_vs_r6.jsonl
python3 tools/ab_inproc.py --shape 1,24,4096,128 --out fp32 --graph --quant 2 --rounds 8 r5=$R r6=$L 2>&1 | grep shape | tee -a $O/ab_r5_vs_r6.jsonl
python3 tools/ab_inproc.py --shape 1,16,8192,128 --out fp32 --graph --quant 2 --rounds 8 r5=$R r6=$L 2>&1 | grep shape | tee -a $O/ab_r5_vs_r6.jsonl
python3 tools/ab_inproc.py --shape 1,16,8192,128 --out fp32 --graph --quant 3 --rounds 8 r5=$R r6=$L 2>&1 | grep shape | tee -a $O/ab_r5_vs_r6.jsonl
