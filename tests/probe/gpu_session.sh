mkdir -p gpurun_out
T=r04zy
for v in "" _oldval; do echo "== lib$v"; DEDF_LIB=diffusion_edf_amd/csrc/libdedf$v.so DEDF_STRESS_LMAX3=1 python tests/stress_parity.py 13 92 sample 2>&1 | grep "sample  12\|sample   3 \|ALL\|FAIL"; done > gpurun_out/${T}_sample_case12_ab.log
cat gpurun_out/${T}_sample_case12_ab.log
