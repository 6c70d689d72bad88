# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
for i in 1 2 3; do python tests/probe/edge_time_sample_fixed.py 2>/dev/null | tail -1; DEDF_LIB=diffusion_edf_amd/csrc/libdedf_norec.so python tests/probe/edge_time_sample_fixed.py 2>/dev/null | tail -1; done | tee gpurun_out/r03t_no_record_stores_timing.log
