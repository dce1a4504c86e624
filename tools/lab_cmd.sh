cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph > /dev/null 2>&1
grep -i "quantize\|w64_i8" gpurun_out/q_stats/runc/*_kernel_stats.csv | cut -c1-200
