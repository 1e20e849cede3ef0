#!/bin/bash
cd $GRAFT_REPO_ROOT
for L in 0 55000 81000; do
echo "SW1_LDS=$L"; if [ $L != 0 ]; then export EDGL_SW1_LDS=$L; fi
EDGL_LABEL_EARLY=1 KT_LINES=14 bash tools/ktrace.sh | cut -c1-150 | grep -i "sweep1"
done
