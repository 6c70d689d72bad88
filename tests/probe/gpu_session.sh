mkdir -p gpurun_out
T=r04v
timeout 2400 python -m pytest tests -m gpu -x -q -s > gpurun_out/${T}_gpu_suite_verbose.log 2>&1
grep -E "TOLPROBE|passed|failed" gpurun_out/${T}_gpu_suite_verbose.log | sort | uniq -c | sort -k2 | tail -150
