# k_plan v2 (atomics-based look-back, rows through LDS): GPU suite, bench, timeline
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02h; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -3 $O/pytest.log | tr '\n' ' ')"
for m in 12 3; do
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
timeout 120 python tools/plan_timeline.py 2>&1 | tail -30
