#!/bin/bash
python3 tools/lab/bias_dbg3.py 4 bf16 2>&1 | grep -v amdgpu
