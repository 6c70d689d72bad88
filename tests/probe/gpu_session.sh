# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r03t_gpu_suite.log; tail -3 gpurun_out/r03t_gpu_suite.log
python bench.py > gpurun_out/r03t_bench.json 2> gpurun_out/r03t_bench.err
python bench.py --lmax 3 > gpurun_out/r03t_lmax3_bench.json 2> gpurun_out/r03t_lmax3_bench.err
python tests/probe/unet_time.py 16384 5 2>&1 | grep -v amdgpu.ids > gpurun_out/r03t_unet_time.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03t_smoke.log 2>&1; tail -2 gpurun_out/r03t_smoke.log
python -c "
import json
for f in ['gpurun_out/r03t_bench.json','gpurun_out/r03t_lmax3_bench.json']:
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline'].get('frac_mfma_issued'), d.get('cpu_baseline',{}).get('value'), d['config'].get('small_batches_50_steps'))"
cat gpurun_out/r03t_unet_time.log
