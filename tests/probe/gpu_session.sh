mkdir -p gpurun_out
T=r04q
python tests/probe/unet_torchprof.py 16384 panda_lowres_lmax3 > gpurun_out/${T}_unet_torchprof.log 2>&1
grep -v "^\[W\|amdgpu.ids\|_warn_once\|UserWarning" gpurun_out/${T}_unet_torchprof.log | head -60 | cut -c1-175
