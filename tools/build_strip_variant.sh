#!/bin/bash
# Builds tools/variants/lib_<name>.so: the library with k_score_strip.hip compiled with extra definitions (timing stamps, ablations).
# usage: bash tools/build_strip_variant.sh <name> "<-D...>"     then     EDGL_LIB_PATH=tools/variants/lib_<name>.so python tools/strip_probe.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/tools/variants"
cd "$ROOT/easydgl_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -fno-slp-vectorize $2 -c k_score_strip.hip -o /tmp/strip_$1.o
OBJS=$(ls obj/*.o | grep -v "obj/k_score_strip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/variants/lib_$1.so" $OBJS /tmp/strip_$1.o
echo "$ROOT/tools/variants/lib_$1.so"
