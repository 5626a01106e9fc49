#!/bin/bash
# the round's last GPU call (a few box-minutes were left): the full GPU suite on the build that ships — the gate for the ASYNC table-copy
# form and the new many-tile tests, neither of which had met the hardware — then the copy leg of bench.py (synchronous and ASYNC rates
# of the same call), then smoke()
TAG=${1:-r05ab}
O=gpurun_out/$TAG; mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD
( time timeout 170 python -m pytest tests -m gpu -q -x ) > $O/tests.log 2>&1 < /dev/null; echo "rc=$?" >> $O/tests.log; tail -n 6 $O/tests.log
( time timeout 60 python bench.py --legs copy --no-cpu-baseline --steps 2 --inner 100 --warmup 1 > $O/bench_copy.json 2> $O/bench_copy.err < /dev/null ); tail -c 1500 $O/bench_copy.json
timeout 40 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 < /dev/null; tail -n 1 $O/smoke.log
