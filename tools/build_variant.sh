#!/bin/bash
# Builds variants/lib_<name>.so: the library with ONE translation unit compiled with extra flags (timing experiments).
# usage: bash tools/build_variant.sh <name> <file without .hip> "<extra flags>"
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N=$1; F=$2; X=$3
mkdir -p "$ROOT/variants"
cd "$ROOT/easydgl_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 $X -c $F.hip -o /tmp/${F}_$N.o
OBJS=$(ls obj/*.o | grep -v "obj/$F.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/variants/lib_$N.so" $OBJS /tmp/${F}_$N.o
echo "$ROOT/variants/lib_$N.so"
