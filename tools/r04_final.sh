#!/bin/bash
# end-of-round evidence for the shipped build: full GPU suite, smoke, HBM traffic passes of k_plan (cfg2), the default bench line,
# rocprofv3 kernel stats of the bench command (cfg2 headline; cfg3 workload). Everything under gpurun_out/r04ba; summaries -> profiles/.
O=gpurun_out/r04ba; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > $O/tests.log 2>&1; tail -4 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/traffic.sh cfg2 > $O/traffic_cfg2.log 2>&1
cp gpurun_out/traffic_cfg2.json $O/ 2>/dev/null
cp gpurun_out/traffic_cfg2.json profiles/ 2>/dev/null   # bench.py reads it from profiles/
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
for wl in cfg2 cfg3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o p -- python bench.py --workload $wl --steps 6 --warmup 1 --inner 100 --legs= --no-cpu-baseline > $O/bench_${wl}_under_rocprof.json 2> $O/prof_$wl.err
  f=$(ls $O/prof_$wl/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && { cp $f $O/${wl}_kernel_stats.csv; head -4 $f | cut -c1-200; }
  t=$(ls $O/prof_$wl/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$t" ] && head -40 $t > $O/${wl}_kernel_trace_head.csv
  rm -rf $O/prof_$wl
done
rm -rf gpurun_out/traffic_cfg2_*
