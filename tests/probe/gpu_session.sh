# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
run() { DEDF_LIB=$1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extractors --no-small-batches 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', round(d['value']), d['roofline']['avg_launch_ms'], round(d['ms_per_step'],4))"; }
for i in 1 2 3; do run diffusion_edf_amd/csrc/libdedf.so base; run diffusion_edf_amd/csrc/libdedf_pda2l2.so pda2; run diffusion_edf_amd/csrc/libdedf_vpda1.so vpda1; done 2>&1 | tee gpurun_out/r03t_ring_depth_lmax2_ab.log
( DEDF_STRESS_LMAX3=1 python tests/stress_parity.py 90 51; DEDF_STRESS_LMAX3=1 python tests/stress_parity.py 30 52 sample; DEDF_STRESS_LMAX3=1 python tests/stress_parity.py 30 53 ebm ) > gpurun_out/r03t_stress.log 2>&1
grep -c "err" gpurun_out/r03t_stress.log; grep "ALL OK\|FAIL\|Traceback" gpurun_out/r03t_stress.log
