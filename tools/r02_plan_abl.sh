# Where does k_plan's time go? Early-exit ablations (ETLG_PLAN_DBG bits, results are wrong: --no-check), one bench each.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
for d in 0 2 4 8 16 24; do
  ETLG_PLAN_DBG=$d timeout 120 python bench.py --workload cfg2 --steps 40 --warmup 5 --pool 4 --no-cpu-baseline --no-scan-leg --no-check > $O/abl_$d.json 2> $O/abl_$d.err
  python - $O/abl_$d.json $d <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
    print(f"dbg {sys.argv[2]:>3}  value {j['value']:8.1f} GB/s  ms/step {j['ms_per_step']:.4f}  {r['kernel']} {r['kernel_avg_us']:.1f} us")
except Exception as e:
    print("FAILED", e, open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
done
