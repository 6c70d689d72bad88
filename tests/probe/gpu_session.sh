# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
DEDF_LIB=diffusion_edf_amd/csrc/libdedf_nodehoist.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "score_parity or sampler_parity or full_size_c2_anchored or half_precision_mode or randomised" 2>&1 | tail -4
run() { DEDF_LIB=diffusion_edf_amd/csrc/libdedf$1.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extractors 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print('lib$1', round(d['value']), round(d['ms_per_step'],4), 'node', round(k['node'],4), d['config']['small_batches_50_steps']['16 poses']['ms_per_step'])"; }
for i in 1 2 3; do run ""; run _nodehoist; done 2>&1 | tee gpurun_out/r03u_node_hoist_ab.log
