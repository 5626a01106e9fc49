cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/parity_all.log 2>&1; echo "parity rc=$?" 
tail -3 gpurun_out/parity_all.log
for x in 0 64 128 256 1024 1984; do echo "DBG_EXTRA=$x"; DBG_EXTRA=$x timeout 120 python tools/dbgt3.py cfg3 2>&1 | tail -1 | cut -c60-400; done
timeout 300 python bench.py --workload cfg3 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120
