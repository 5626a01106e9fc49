#!/bin/bash
# evidence for a build that differs from the one tools/r05_final.sh measured only in the hand-off calls: the full GPU suite, smoke, the
# default bench line (traffic / rocprof kernel stats of the decode kernels: the earlier run's stand)
TAG=${1:-r05zz}
O=gpurun_out/$TAG; mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD
( time timeout 600 python -m pytest tests -m gpu -q -x ) > $O/tests.log 2>&1 < /dev/null; tail -n 4 $O/tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 < /dev/null; tail -n 1 $O/smoke.log
( time timeout 600 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null ); tail -c 300 $O/bench.json
