# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
T=r04a
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "one_time or workspace or anchored or overflow" > gpurun_out/${T}_tests_parity.log 2>&1; tail -5 gpurun_out/${T}_tests_parity.log
python -m pytest tests/test_lmax3.py -m gpu -q -k "half_precision_mode_lmax3 or anchored" > gpurun_out/${T}_tests_lmax3.log 2>&1; tail -5 gpurun_out/${T}_tests_lmax3.log
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; tail -c 600 gpurun_out/${T}_bench.json
python bench.py --config5 > gpurun_out/${T}_config5_bench.json 2> gpurun_out/${T}_config5_bench.err; tail -c 1500 gpurun_out/${T}_config5_bench.json; tail -5 gpurun_out/${T}_config5_bench.err
python -m pytest tests/test_config5.py -m gpu -q -s > gpurun_out/${T}_tests_config5.log 2>&1; tail -8 gpurun_out/${T}_tests_config5.log
