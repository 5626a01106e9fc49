#!/bin/bash
# end-of-round evidence for the shipped build: full GPU suite, smoke, HBM traffic passes (cfg2, cfg3), the default bench line,
# rocprofv3 kernel stats of the bench command (cfg2 headline; cfg3 workload). Everything under gpurun_out/$TAG; summaries -> profiles/.
TAG=${1:-r05m}
O=gpurun_out/$TAG; mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/tests.log 2>&1; tail -n 4 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for wl in cfg2 cfg3; do
  bash tools/traffic.sh $wl > $O/traffic_$wl.log 2>&1
  cp gpurun_out/traffic_$wl.json $O/ 2>/dev/null
  cp gpurun_out/traffic_$wl.json profiles/ 2>/dev/null   # bench.py reads it from profiles/
done
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ); tail -c 300 $O/bench.json
for wl in cfg2 cfg3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o p -- python bench.py --workload $wl --steps 6 --warmup 1 --inner 100 --legs= --no-cpu-baseline > $O/bench_${wl}_under_rocprof.json 2> $O/prof_$wl.err
  f=$(ls $O/prof_$wl/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && { cp $f $O/${wl}_kernel_stats.csv; head -n 4 $f | cut -c1-200; }
  t=$(ls $O/prof_$wl/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$t" ] && head -n 40 $t > $O/${wl}_kernel_trace_head.csv
  rm -rf $O/prof_$wl
done
rm -rf gpurun_out/traffic_cfg2_* gpurun_out/traffic_cfg3_*
