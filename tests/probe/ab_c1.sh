#!/bin/bash
# A/B of two library builds on config C1 (lmax 1, 2k/512, 256 poses x 50 steps), where the small kernels weigh most
run() { DEDF_LIB=$1 python bench.py --lmax 1 --scene 2048 --grasp 512 --poses-per-gpu 256 --steps 50 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', round(d['value']), round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['roofline']['kernel_ms_per_step'].items()})"; }
for i in 1 2; do run diffusion_edf_amd/csrc/libdedf_base.so base; run diffusion_edf_amd/csrc/libdedf.so var; done
