#!/bin/bash
set -x
O=gpurun_out/r02n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_async.py tests/test_gpu_scan.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 300 python bench.py --steps 3 --warmup 1 --prime 2 --legs no_sidecar,handoff --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02n/bench.json").read().strip().split("\n")[-1])
print(d["value"], d["no_sidecar"])
PY
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python bench.py --steps 2 --warmup 1 --prime 2 --legs no_sidecar,handoff --no-cpu-baseline > /dev/null 2> $O/prof.err
f=$(ls $O/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && { cp $f $O/kernel_stats.csv; head -30 $f; }
rm -rf $O/prof
