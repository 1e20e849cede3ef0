#!/bin/bash
cd $GRAFT_REPO_ROOT
export EDGL_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_nosb.so
EDGL_LABEL_EARLY=1 KT_LINES=14 bash tools/ktrace.sh | cut -c1-150 | grep -i "sweep"
unset EDGL_LIB_PATH
EDGL_LABEL_EARLY=1 KT_LINES=14 bash tools/ktrace.sh | cut -c1-150 | grep -i "sweep"
