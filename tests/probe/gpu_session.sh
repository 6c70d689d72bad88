# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
run() { DEDF_LIB=diffusion_edf_amd/csrc/libdedf$1.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extractors --no-small-batches 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib$1', round(d['value']), d['roofline']['avg_launch_ms'], round(d['ms_per_step'],4))"; }
for i in 1 2 3; do run ""; run _maxilp; run _trackers; run _nopost; run _memclause; done 2>&1 | tee gpurun_out/r03t_sched_flags_lmax2_ab.log
