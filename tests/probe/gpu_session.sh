# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
for P in 1000 16; do for i in 1 2; do for v in "" _nodewsame _nodewnone; do POSES=$P DEDF_LIB=diffusion_edf_amd/csrc/libdedf$v.so python tests/probe/edge_time_sample_fixed.py 2>/dev/null | tail -1 | sed "s/^/poses $P: /"; done; done; done | tee gpurun_out/r03u_node_weight_path_timing.log
