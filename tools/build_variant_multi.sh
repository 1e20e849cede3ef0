#!/bin/bash
# Builds variants/lib_<name>.so: the library with SEVERAL translation units rebuilt under extra flags (timing experiments, A/B
# through EDGL_LIB_PATH on one box).  usage: bash tools/build_variant_multi.sh <name> "<extra flags>" <file without .hip> ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N=$1; X=$2; shift 2
mkdir -p "$ROOT/tools/variants"
cd "$ROOT/easydgl_amd/csrc"
SKIP=""; NEW=""
for F in "$@"; do
  EXTRA=""
  case $F in k_bimau_*|k_tattn) EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1";; k_score_strip) EXTRA="-fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value $EXTRA $X -c $F.hip -o /tmp/${F}_$N.o &
  SKIP="$SKIP -e obj/$F.o"; NEW="$NEW /tmp/${F}_$N.o"
done
wait
OBJS=$(ls obj/*.o | grep -v -x $SKIP)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/variants/lib_$N.so" $OBJS $NEW
echo "$ROOT/tools/variants/lib_$N.so"
