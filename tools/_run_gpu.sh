#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/pytest_gpu.txt
cat gpurun_out/pytest_gpu.txt
for v in base nohash skip1; do
  if [ $v = base ]; then unset EDGL_LIB_PATH; else export EDGL_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/lib_$v.so; fi
  KT_LINES=14 bash tools/ktrace.sh > gpurun_out/ab_$v.txt 2>&1
  echo "== $v"; grep -E "bimau|intensity|metric" gpurun_out/ab_$v.txt | cut -c1-200
done
unset EDGL_LIB_PATH
python bench.py > gpurun_out/bench_default.log 2>&1
tail -1 gpurun_out/bench_default.log | cut -c1-300
