# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r03v_gpu_suite.log; tail -3 gpurun_out/r03v_gpu_suite.log
python bench.py > gpurun_out/r03v_bench.json 2> gpurun_out/r03v_bench.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03v_smoke.log 2>&1; tail -2 gpurun_out/r03v_smoke.log
python -c "
import json
d=json.loads(open('gpurun_out/r03v_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline'].get('kernel_ms_per_step'))"
