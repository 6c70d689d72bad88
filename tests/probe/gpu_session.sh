# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
python -m pytest tests/test_lmax3.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r03q_test_lmax3.log; tail -3 gpurun_out/r03q_test_lmax3.log
python bench.py --lmax 3 --no-cpu-baseline > gpurun_out/r03q_lmax3_bench.json 2> gpurun_out/r03q_lmax3_bench.err
python -c "
import json
d=json.loads(open('gpurun_out/r03q_lmax3_bench.json').read().strip().splitlines()[-1]); print('lmax3', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config'].get('score_fwd_ms_at_t0.5'), d['config'].get('feature_extractors_ms'), d['config'].get('small_batches_50_steps'))"
