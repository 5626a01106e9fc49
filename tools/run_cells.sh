cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_scan.py -m gpu -x -q > gpurun_out/scan.log 2>&1; echo "scan rc=$?"; tail -15 gpurun_out/scan.log
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/parity_all.log 2>&1; echo "parity rc=$?" 
tail -3 gpurun_out/parity_all.log
