# k_plan on the GPU: full GPU suite, then bench + rocprofv3 --kernel-trace --stats on cfg2 (sweeping the LDS margin = occupancy)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -3 $O/pytest.log | tr '\n' ' ')"
for m in 12 3 40; do
  ETLG_PLAN_MARGIN=$m timeout 200 python bench.py --workload cfg2 --steps 60 --warmup 5 --no-cpu-baseline --no-scan-leg > $O/bench_cfg2_m$m.json 2> $O/bench_cfg2_m$m.err
  python - $O/bench_cfg2_m$m.json $m <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
    print(f"margin {sys.argv[2]:>3}%  value {j['value']:8.1f} GB/s  ms/step {j['ms_per_step']:.4f}  {r['kernel']} {r['kernel_avg_us']:.1f} us  frac {r['frac']}  read_frac {j['hbm_read_frac']}")
except Exception as e:
    print("FAILED", e, open(sys.argv[1].replace('.json', '.err')).read()[-800:])
PY
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2 -o p -- python bench.py --workload cfg2 --steps 60 --warmup 5 --no-cpu-baseline --no-scan-leg > $O/bench_cfg2_under_rocprof.json 2> $O/prof_cfg2.err
f=$(ls $O/prof_cfg2/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && { cp $f $O/cfg2_kernel_stats.csv; head -5 $f; }
t=$(ls $O/prof_cfg2/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$t" ] && head -40 $t > $O/cfg2_kernel_trace_head.csv
rm -rf $O/prof_cfg2
