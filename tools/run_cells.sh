cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "float" > gpurun_out/float.log 2>&1; echo "float rc=$?"; tail -15 gpurun_out/float.log
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/parity_all.log 2>&1; echo "parity rc=$?" 
tail -3 gpurun_out/parity_all.log
