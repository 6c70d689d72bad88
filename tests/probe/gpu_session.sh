# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
python -m pytest tests/test_unet.py -m gpu -x -q -k "shared_workspace" 2>&1 | tail -5 > gpurun_out/r03o_test_unet.log
tail -3 gpurun_out/r03o_test_unet.log
run() { DEDF_LIB=$1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extractors 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', round(d['value']), d['roofline']['avg_launch_ms'], round(d['ms_per_step'],4))"; }
for i in 1 2; do run diffusion_edf_amd/csrc/libdedf.so base; run diffusion_edf_amd/csrc/libdedf_pairl2.so pairl2; done 2>&1 | tee gpurun_out/r03o_pairl2_ab.log
LMAX=3 SAMPLE=1 DEDF_LIB=diffusion_edf_amd/csrc/libdedf_prof3.so python tests/phase_prof.py 2>&1 | tail -22 | tee gpurun_out/r03o_phase_lmax3.log
