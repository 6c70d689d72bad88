#!/bin/bash
run() { DEDF_LIB=$1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', round(d['value']), d['roofline']['kernel_ms_per_step']['edge'], d['roofline']['kernel_ms_per_step']['node'], round(d['roofline']['kernel_ms_per_step']['aggregate'],4), round(d['roofline']['kernel_ms_per_step']['neighbors'],4))"; }
for i in 1 2; do run diffusion_edf_amd/csrc/libdedf_base.so base; run diffusion_edf_amd/csrc/libdedf.so var; done
