cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02l; mkdir -p $O
for d in 0 128; do
  ETLG_PLAN_DBG=$d timeout 200 python bench.py --workload cfg2 --steps 60 --warmup 5 --no-cpu-baseline --no-scan-leg > $O/wt_$d.json 2> $O/wt_$d.err
  python - $O/wt_$d.json $d <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
    print(f"dbg {sys.argv[2]:>5}  value {j['value']:8.1f} GB/s  ms/step {j['ms_per_step']:.4f}  {r['kernel']} {r['kernel_avg_us']:.1f} us  frac {r['frac']}  read_frac {j['hbm_read_frac']}")
except Exception as e:
    print("FAILED", e, open(sys.argv[1].replace('.json', '.err')).read()[-800:])
PY
done
