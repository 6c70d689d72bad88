mkdir -p gpurun_out
T=r04n
( for i in 1 2; do
  DEDF_LIB=diffusion_edf_amd/csrc/libdedf.so python tests/probe/edge_time_sample_fixed.py 2>&1 | tail -1
  for v in tA tS1 tS2; do DEDF_LIB=diffusion_edf_amd/csrc/libdedf_$v.so python tests/probe/edge_time_sample_fixed.py 2>&1 | tail -1; done
  for v in tA2 tS1w2 tS2w2; do DEDF_EDGE_WAVES_PER_CU=8 DEDF_LIB=diffusion_edf_amd/csrc/libdedf_$v.so python tests/probe/edge_time_sample_fixed.py 2>&1 | tail -1; done
  DEDF_EDGE_WAVES_PER_CU=8 DEDF_LIB=diffusion_edf_amd/csrc/libdedf_tS2.so python tests/probe/edge_time_sample_fixed.py 2>&1 | tail -1
done ) > gpurun_out/${T}_two_waves_per_simd_timing.log 2>&1
cat gpurun_out/${T}_two_waves_per_simd_timing.log
