#!/bin/bash
# A/B of several library builds on C2: bash tests/probe/ab_multi.sh libA.so libB.so ...   (paths relative to diffusion_edf_amd/csrc)
run() { DEDF_LIB=diffusion_edf_amd/csrc/$1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['roofline']['kernel_ms_per_step']['edge'],4), round(d['ms_per_step'],4))"; }
for i in 1 2; do for l in "$@"; do run $l; done; done
