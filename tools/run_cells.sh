# parity of the column-parallel kernel on every scenario + cfg3 timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
ETLG_FUSED_KERNEL=2 timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/cells_parity.log 2>&1; echo "cells parity rc=$?" 
tail -5 gpurun_out/cells_parity.log
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/default_parity.log 2>&1; echo "default parity rc=$?"
tail -3 gpurun_out/default_parity.log
timeout 300 python bench.py --workload cfg3 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/cfg3_cells.log 2>&1; tail -2 gpurun_out/cfg3_cells.log
ETLG_FUSED_KERNEL=1 timeout 300 python bench.py --workload cfg3 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/cfg3_f64.log 2>&1; tail -1 gpurun_out/cfg3_f64.log
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/cfg2.log 2>&1; tail -1 gpurun_out/cfg2.log
