# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
T=r04j
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sampler or radial or table or c2 or c1 or c3 or overflow or workspace or philox or tiny or zero_edge" > gpurun_out/${T}_tests_sampler.log 2>&1; tail -4 gpurun_out/${T}_tests_sampler.log
python -m pytest tests/test_lmax3.py -m gpu -q -x -k "sampler" > gpurun_out/${T}_tests_lmax3.log 2>&1; tail -3 gpurun_out/${T}_tests_lmax3.log
python tests/probe/small_batch.py 2 200 2>&1 | grep lmax > gpurun_out/${T}_small_batch.log; cat gpurun_out/${T}_small_batch.log
DEDF_RTAB_ASYNC=0 python tests/probe/small_batch.py 2 200 2>&1 | grep lmax > gpurun_out/${T}_small_batch_sync_tables.log; cat gpurun_out/${T}_small_batch_sync_tables.log
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extractors > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print('async ', round(d['value']), round(d['ms_per_step'],4), 'edge', round(r['avg_launch_ms'],4), r['kernel_ms_per_step'], d['config']['small_batches_50_steps'])"
DEDF_RTAB_ASYNC=0 python bench.py --no-cpu-baseline --no-extractors > gpurun_out/${T}_bench_sync.json 2> gpurun_out/${T}_bench_sync.err; python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_sync.json').read().strip().splitlines()[-1]); r=d['roofline']; print('sync  ', round(d['value']), round(d['ms_per_step'],4), 'edge', round(r['avg_launch_ms'],4), r['kernel_ms_per_step'], d['config']['small_batches_50_steps'])"
done
