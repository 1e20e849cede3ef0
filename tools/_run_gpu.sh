cd $GRAFT_REPO_ROOT
KT_LINES=45 bash tools/ktrace.sh 2>&1 | tail -50
