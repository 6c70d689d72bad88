mkdir -p gpurun_out
T=r04w
for r in 1 2 3; do for v in _base ""; do DEDF_LIB=diffusion_edf_amd/csrc/libdedf$v.so python tests/probe/edge_time_sample_fixed.py 2>/dev/null | tail -1; done; done > gpurun_out/${T}_chain0_ab.log
for r in 1 2; do for v in _base ""; do LMAX=3 DEDF_LIB=diffusion_edf_amd/csrc/libdedf$v.so python tests/probe/edge_time_sample_fixed.py 2>/dev/null | tail -1; done; done > gpurun_out/${T}_chain0_lmax3_ab.log
cat gpurun_out/${T}_chain0_ab.log gpurun_out/${T}_chain0_lmax3_ab.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
