cd $GRAFT_REPO_ROOT
for x in 0 2048 4096 6144 8192; do echo "DBG_EXTRA=$x"; DBG_EXTRA=$x timeout 120 python tools/dbgt3.py cfg3 2>&1 | tail -1 | cut -c60-400; done
