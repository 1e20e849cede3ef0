#!/bin/bash
cd $GRAFT_REPO_ROOT
EDGL_LABEL_EARLY=1 KT_LINES=14 bash tools/ktrace.sh | cut -c1-150 | grep -i "sweep"
EDGL_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_dbranch.so EDGL_LABEL_EARLY=1 KT_LINES=14 bash tools/ktrace.sh | cut -c1-150 | grep -i "sweep"
