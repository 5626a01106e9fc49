cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/parity_all.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/parity_all.log
timeout 300 python tools/bench_copy.py 400000 2>&1 | tail -1
