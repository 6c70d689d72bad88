# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
for i in 1 2 3; do for v in _noanchor _anchor; do DEDF_LIB=diffusion_edf_amd/csrc/libdedf$v.so python tests/probe/edge_time_sample_fixed.py 2>/dev/null | tail -1; done; done | tee gpurun_out/r03t_anchor_timing.log
