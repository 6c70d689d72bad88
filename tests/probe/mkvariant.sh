#!/bin/bash
# Build a variant of libdedf.so in which kernel unit $UNIT (default 0 = k_edge<2,128,false>, the headline instantiation) is compiled with extra flags:
#   bash tests/probe/mkvariant.sh <name> [extra hipcc flags...]   ->  diffusion_edf_amd/csrc/libdedf_<name>.so
# The other units are taken from the last regular build (diffusion_edf_amd/csrc/_obj); with API=1 in the environment the API object
# is rebuilt with the same extra flags too (needed for -DDEDF_PHASE_PROF).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
C=$ROOT/diffusion_edf_amd/csrc
NAME=$1; shift
UNIT=${UNIT:-0}      # UNIT=12: the sampler's table-reading instantiation
F="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-slp-vectorize"
hipcc $F -c -DDEDF_KUNIT=$UNIT "$@" $C/dedf_kernels.hip -o $C/_obj/k${UNIT}_$NAME.o &
APIO=$C/_obj/api.o
if [ -n "$API" ]; then APIO=$C/_obj/api_$NAME.o; hipcc $F -c "$@" $C/dedf_api.hip -o $APIO & fi
wait
OBJS="$APIO $C/_obj/k${UNIT}_$NAME.o"
NU=$(grep -o "kKernelUnits = [0-9]*" $C/dedf_kernel_list.h | grep -o "[0-9]*$")
for u in $(seq 0 $((NU - 1))); do if [ $u != $UNIT ]; then OBJS="$OBJS $C/_obj/k$u.o"; fi; done

hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $C/libdedf_$NAME.so
echo built $C/libdedf_$NAME.so
