#!/bin/bash
# round 4, trip A: the default bf16 forward with the P V product in fp16 (V converted in the kernel): parity, then A/B against
# the bf16 P V kernels; in-kernel clock stamps of the headline kernel; kernel-trace + SQ PMC of BASELINE config 2 (evidence asked
# for by the round-3 review: none existed for fa_fwd16<bf16,64> at B4 H16 S1024 causal)
O=gpurun_out/r4a; mkdir -p $O
export TMPDIR=/tmp
timeout 1700 python -m pytest tests/test_gpu_w64.py tests/test_gpu_forward.py tests/test_gpu_configs.py tests/test_gpu_rope_fused.py tests/test_gpu_sdpa.py -x -q > $O/tests.txt 2>&1; tail -15 $O/tests.txt
timeout 900 python tools/lab/pv16_probe.py > $O/pv16_probe.jsonl 2>$O/probe_err.txt; cat $O/pv16_probe.jsonl; tail -3 $O/probe_err.txt
UMFA_LIBRARY=$PWD/tools/lab_bin/libMFAFFI_stamps.so timeout 300 python tools/w64_stamps.py 1 24 4096 128 > $O/stamps_flux.txt 2>&1; cat $O/stamps_flux.txt
UMFA_LIBRARY=$PWD/tools/lab_bin/libMFAFFI_stamps.so UMFA_PV_FP16=0 timeout 300 python tools/w64_stamps.py 1 24 4096 128 > $O/stamps_flux_pv0.txt 2>&1; cat $O/stamps_flux_pv0.txt
# config 2: kernel trace + SQ counters
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg2_trace -- python3 tools/run_fwd.py 200 4 16 1024 64 causal > $O/cfg2_trace.txt 2>$O/prof_err.txt
find $O/cfg2_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/cfg2_kernel_stats.csv; cut -c1-220 $O/cfg2_kernel_stats.csv | head -5
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_cfg2 -- python3 tools/run_fwd.py 20 4 16 1024 64 causal > /dev/null 2>>$O/prof_err.txt
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_cfg2b -- python3 tools/run_fwd.py 20 4 16 1024 64 causal > /dev/null 2>>$O/prof_err.txt
python3 tools/pmc_summary.py $O/pmc_cfg2 $O/pmc_cfg2b > $O/pmc_cfg2_summary.txt 2>&1; cat $O/pmc_cfg2_summary.txt
find $O -name "*.db" -delete; find $O -type d -name "cfg2_trace" -exec rm -rf {} + 2>/dev/null; tail -3 $O/prof_err.txt
