cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/parity_all.log 2>&1; echo "parity rc=$?" 
tail -3 gpurun_out/parity_all.log
timeout 120 python tools/dbgt3.py cfg3 2>&1 | tail -1 | cut -c60-500
timeout 300 python bench.py --workload cfg3 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120
