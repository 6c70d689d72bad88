mkdir -p gpurun_out
T=r04x
for r in 1 2 3 4; do for v in _base _c0only ""; do DEDF_LIB=diffusion_edf_amd/csrc/libdedf$v.so python tests/probe/edge_time_sample_fixed.py 2>/dev/null | tail -1; done; done > gpurun_out/${T}_chain_variants_ab.log
for r in 1 2 3; do for v in _base _c0only3 ""; do LMAX=3 DEDF_LIB=diffusion_edf_amd/csrc/libdedf$v.so python tests/probe/edge_time_sample_fixed.py 2>/dev/null | tail -1; done; done > gpurun_out/${T}_chain_variants_lmax3_ab.log
cat gpurun_out/${T}_chain_variants_ab.log gpurun_out/${T}_chain_variants_lmax3_ab.log | awk '{print $1, $4}'
