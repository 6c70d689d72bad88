#!/bin/bash
# scratch driver for one gpurun call (rewritten per session)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu -k "half" > gpurun_out/r05v_half_tests.log 2>&1; tail -3 gpurun_out/r05v_half_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 --half > gpurun_out/r05v_half_bench.json 2> gpurun_out/r05v_half_bench.err; tail -c 600 gpurun_out/r05v_half_bench.json
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05v_bench.json 2> gpurun_out/r05v_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 --half --lmax 3 > gpurun_out/r05v_half_lmax3_bench.json 2>&1
python -m pytest tests -x -q -m gpu --durations=5 > gpurun_out/r05v_gpu_suite.log 2>&1; tail -3 gpurun_out/r05v_gpu_suite.log
