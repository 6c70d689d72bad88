# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r03u_gpu_suite.log; tail -3 gpurun_out/r03u_gpu_suite.log
python bench.py > gpurun_out/r03u_bench.json 2> gpurun_out/r03u_bench.err
python bench.py --lmax 3 > gpurun_out/r03u_lmax3_bench.json 2> gpurun_out/r03u_lmax3_bench.err
python bench.py --lmax 1 --scene 2048 --grasp 512 --poses-per-gpu 256 --steps 50 --no-cpu-baseline > gpurun_out/r03u_c1_bench.json 2> gpurun_out/r03u_c1_bench.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03u_smoke.log 2>&1; tail -2 gpurun_out/r03u_smoke.log
python tests/probe/small_batch.py > gpurun_out/r03u_small_batch.log 2>&1; tail -5 gpurun_out/r03u_small_batch.log
python -c "
import json
for f in ['gpurun_out/r03u_bench.json','gpurun_out/r03u_lmax3_bench.json','gpurun_out/r03u_c1_bench.json']:
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline'].get('kernel_ms_per_step'), d['config'].get('small_batches_50_steps'))"
