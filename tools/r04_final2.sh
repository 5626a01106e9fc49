#!/bin/bash
# the round's last build: full GPU suite, smoke, the default bench line -> gpurun_out/r04bb (kernel stats / traffic of the decode kernels: r04ba, unchanged since)
O=gpurun_out/r04bb; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
