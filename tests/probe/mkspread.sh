#!/bin/bash
# Variant of libdedf.so in which kernel unit $UNIT went through tests/probe/mfma_spread.py (post-pass over hipcc's assembly):
#   UNIT=33 GAP=6 bash tests/probe/mkspread.sh <name> [extra hipcc flags...]   ->  diffusion_edf_amd/csrc/libdedf_<name>.so
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
C=$ROOT/diffusion_edf_amd/csrc
L=/opt/rocm/lib/llvm/bin
NAME=$1; shift
UNIT=${UNIT:-33}; GAP=${GAP:-6}
F="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-slp-vectorize"
T=$(mktemp -d)
hipcc $F -DDEDF_KUNIT=$UNIT "$@" -S --cuda-device-only $C/dedf_kernels.hip -o $T/u.s 2>/dev/null
python3 $ROOT/tests/probe/mfma_spread.py $T/u.s $T/u_sp.s --gap $GAP --report
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/u_sp.s -o $T/u_dev.o
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared $T/u_dev.o -o $T/u.hsaco
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/u.hsaco -output=$T/u.hipfb
hipcc $F -DDEDF_KUNIT=$UNIT "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/u.hipfb -c $C/dedf_kernels.hip -o $C/_obj/k${UNIT}_$NAME.o
OBJS="$C/_obj/api.o $C/_obj/k${UNIT}_$NAME.o"
NU=$(grep -o "kKernelUnits = [0-9]*" $C/dedf_kernel_list.h | grep -o "[0-9]*$")
for u in $(seq 0 $((NU - 1))); do if [ $u != $UNIT ]; then OBJS="$OBJS $C/_obj/k$u.o"; fi; done

hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $C/libdedf_$NAME.so
rm -rf $T
echo built $C/libdedf_$NAME.so
