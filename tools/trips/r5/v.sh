#!/bin/bash
# trip v: bounded wait of the cast pass (new test), cast pass timing with it, block-diagonal A/B, the soak legs of the value fuzz
O=gpurun_out/r5v; mkdir -p $O
export TMPDIR=/tmp
python3 -m pytest tests/test_gpu_pv16_range.py tests/test_gpu_w64.py tests/test_gpu_streams.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
cp universal-metal-flash-attention_amd/lib/libMFAFFI.so /tmp/libMFAFFI_nowait.so
python3 tools/ab_inproc.py --shape 1,24,4096,128 --out fp32 --graph r4=tools/lab_bin/libMFAFFI_r4.so new=universal-metal-flash-attention_amd/lib/libMFAFFI.so "nowait=/tmp/libMFAFFI_nowait.so:cast_wait_us=0" > $O/ab_flux.txt 2>&1; tail -4 $O/ab_flux.txt
