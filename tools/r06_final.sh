#!/bin/bash
# end-of-round evidence for the shipped build (round 6): full GPU suite, smoke, HBM traffic passes (separate --pmc runs: cfg2 k_plan, cfg3 / cfg5 /
# wide70 k_rows, copy k_copy_cells, the boundary scan), the default bench line, rocprofv3 kernel stats of the bench command (cfg2
# headline; cfg3 workload) and of one launch series per side leg, the long fuzzers. Everything under gpurun_out/$TAG; summaries -> profiles/.
TAG=${1:-r06f}
O=gpurun_out/$TAG; mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/tests.log 2>&1; tail -n 4 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for wl in cfg2 cfg3; do
  bash tools/traffic.sh $wl > $O/traffic_$wl.log 2>&1
  cp gpurun_out/traffic_$wl.json $O/ 2>/dev/null; cp gpurun_out/traffic_$wl.json profiles/ 2>/dev/null
done
for wl in cfg5 wide70 copy copy_clean; do
  bash tools/traffic.sh $wl python tools/one_kernel.py $wl 4 > $O/traffic_$wl.log 2>&1
  cp gpurun_out/traffic_$wl.json $O/ 2>/dev/null; cp gpurun_out/traffic_$wl.json profiles/ 2>/dev/null
done
TRAFFIC_KERNEL=k_bounds_local bash tools/traffic.sh scan python tools/one_kernel.py nosidecar 4 > $O/traffic_scan.log 2>&1
cp gpurun_out/traffic_scan.json $O/ 2>/dev/null; cp gpurun_out/traffic_scan.json profiles/ 2>/dev/null
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ); tail -c 300 $O/bench.json
for wl in cfg2 cfg3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o p -- python bench.py --workload $wl --steps 6 --warmup 1 --inner 100 --legs= --no-cpu-baseline > $O/bench_${wl}_under_rocprof.json 2> $O/prof_$wl.err
  f=$(ls $O/prof_$wl/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && { cp $f $O/${wl}_kernel_stats.csv; head -n 4 $f | cut -c1-200; }
  t=$(ls $O/prof_$wl/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$t" ] && head -n 40 $t > $O/${wl}_kernel_trace_head.csv
  rm -rf $O/prof_$wl
done
for wl in cfg5 wide70 copy copy_clean nosidecar finish; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o p -- python tools/one_kernel.py $wl 8 > /dev/null 2> $O/prof_$wl.err
  f=$(ls $O/prof_$wl/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && { cp $f $O/${wl}_kernel_stats.csv; head -n 3 $f | cut -c1-160; }
  rm -rf $O/prof_$wl
done
for m in nosidecar mixed cfg2 cfg5 ddl_fixed; do timeout 200 python tools/async_long_fuzz.py 40 11 $m 2>&1 | tail -1 >> $O/long_fuzz.txt; done
timeout 300 python tools/cell_fuzz.py 60 2>&1 | tail -2 >> $O/long_fuzz.txt
rm -rf gpurun_out/traffic_*_FETCH_SIZE gpurun_out/traffic_*_WRITE_SIZE
