cd $GRAFT_REPO_ROOT
timeout 120 python tools/dbgt3.py cfg3 2>&1 | tail -1 | cut -c60-500
