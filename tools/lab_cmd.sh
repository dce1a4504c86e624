#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_quantized.py -m gpu -q -k wide_dynamic 2>&1 | tail -30
UMFA_LIBRARY=tools/lab_bin/libMFAFFI_I2F.so timeout 900 python -m pytest tests/test_gpu_quantized.py -m gpu -q -k wide_dynamic 2>&1 | tail -30
