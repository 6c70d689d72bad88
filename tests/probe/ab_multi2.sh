#!/bin/bash
# A/B several library variants on fixed poses (tests/probe/edge_time.py: k_edge ms at C2, t = 0.5), interleaved, 2 rounds
#   bash tests/probe/ab_multi2.sh base noslp ...
for i in 1 2; do for v in "$@"; do
  if [ "$v" = base ]; then L=diffusion_edf_amd/csrc/libdedf.so; else L=diffusion_edf_amd/csrc/libdedf_$v.so; fi
  DEDF_LIB=$L python tests/probe/edge_time.py 2>&1 | tail -1
done; done
