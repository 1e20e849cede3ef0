#!/bin/bash
# Builds tools/variants/lib_phase_<file>.so: the library with ONE translation unit compiled with -DEDGL_PHASE_TIMING.
# usage: [EXTRA=-DEDGL_PHASE_BWD] bash tools/build_phase_variant.sh k_tail
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
F=$1
mkdir -p "$ROOT/tools/variants"
cd "$ROOT/easydgl_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -DEDGL_PHASE_TIMING $EXTRA -c $F.hip -o /tmp/${F}_phase.o
OBJS=$(ls obj/*.o | grep -v "obj/$F.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/variants/lib_phase_${F}${TAG}.so" $OBJS /tmp/${F}_phase.o
echo "$ROOT/tools/variants/lib_phase_${F}${TAG}.so"
