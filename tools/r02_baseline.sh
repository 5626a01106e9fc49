# Round-2 first GPU call: the shipped (r01j) library under the full GPU suite, rocprofv3 --kernel-trace --stats of the bench
# command on cfg2 and cfg3 (the evidence VERDICT r01 found missing for that build), and the flag variants on k_cells (cfg3).
#   gpurun --timeout 900 -- 'bash tools/r02_baseline.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for wl in cfg2 cfg3; do
  timeout 200 python bench.py --workload $wl --steps 60 --warmup 5 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err
  echo "bench $wl: $(head -c 600 $O/bench_$wl.json)"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o p -- python bench.py --workload $wl --steps 60 --warmup 5 --no-cpu-baseline --no-scan-leg > $O/bench_${wl}_under_rocprof.json 2> $O/prof_$wl.err
  f=$(ls $O/prof_$wl/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && { cp $f $O/${wl}_kernel_stats.csv; head -6 $f; }
  t=$(ls $O/prof_$wl/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$t" ] && head -40 $t > $O/${wl}_kernel_trace_head.csv
  rm -rf $O/prof_$wl
done
for v in plain hotfix early_hotfix stage8 all; do
  lib=$GRAFT_REPO_ROOT/etl_amd/variants/libetl_gfx950_$v.so
  [ -f $lib ] || continue
  ETLG_LIB_PATH=$lib timeout 120 python bench.py --workload cfg3 --steps 40 --warmup 5 --pool 4 --no-cpu-baseline --no-scan-leg > $O/var_${v}_cfg3.json 2> $O/var_${v}_cfg3.err
  python - $v $O <<'PY'
import json, sys
v, O = sys.argv[1:3]
try:
    j = json.loads(open(f"{O}/var_{v}_cfg3.json").read().strip().splitlines()[-1]); r = j["roofline"]
    print(f"{v:16s} cfg3 value {j['value']:8.1f} GB/s  {r['kernel']} {r['kernel_avg_us']:.1f} us  frac {r['frac']}")
except Exception as e:
    print(v, "FAILED", e)
PY
done
