#!/bin/bash
# On the GPU box: A/B of compile-time switches of k_score_strip.hip (the headline width: C = 128, 10240 slots, 20001 items).
# (quote them; "" = the defaults); every variant is built there, linked against the other objects and timed with tools/strip_bench.py.
#   bash tools/stripw_ab.sh "" "-DSTRIPW_SPREAD=0" "-DSTRIPW_PF=7"       env: C R I ZERO as tools/strip_bench.py, REPS (default 2)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT/easydgl_amd/csrc"
export C=${C:-128} R=${R:-10240} I=${I:-20001}
n=0
for defs in "$@"; do
  n=$((n+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -fno-slp-vectorize $defs -c k_score_strip.hip -o /tmp/strip_v$n.o || { echo "variant $n ($defs): build failed"; continue; }
  OBJS=$(ls obj/*.o | grep -v "obj/k_score_strip.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_s$n.so $OBJS /tmp/strip_v$n.o
done
cd "$ROOT"
for rep in $(seq 1 ${REPS:-2}); do
  n=0
  for defs in "$@"; do
    n=$((n+1))
    [ -f /tmp/lib_s$n.so ] || continue
    echo -n "[$defs] "
    EDGL_LIB_PATH=/tmp/lib_s$n.so python tools/strip_bench.py 2>&1 | grep -v amdgpu.ids
  done
done
