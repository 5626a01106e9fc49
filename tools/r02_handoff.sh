#!/bin/bash
# GPU: new columnar hand-off tests + bench leg + kernel stats
set -x
mkdir -p gpurun_out/r02h
timeout 900 python -m pytest tests/test_gpu_columns.py tests/test_gpu_rowbinary.py -m gpu -x -q > gpurun_out/r02h/tests.log 2>&1; tail -5 gpurun_out/r02h/tests.log
timeout 300 python bench.py --steps 3 --warmup 1 --prime 2 --legs handoff --no-cpu-baseline > gpurun_out/r02h/bench.json 2> gpurun_out/r02h/bench.err; tail -c 1500 gpurun_out/r02h/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02h/prof -o h -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --prime 2 --legs handoff --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r02h/prof -name "*kernel_stats*" | head -1 | xargs -I{} head -25 {}
