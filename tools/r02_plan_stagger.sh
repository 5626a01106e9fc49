# k_plan rolling start: sweep the step (ETLG_PLAN_STAGGER, 1/1024 of 64 cycles per tile; 0 = off) and the phase clocks
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e; mkdir -p $O
for st in 0 16 32 48 64 96 128; do
  ETLG_PLAN_STAGGER=$st timeout 120 python bench.py --workload cfg2 --steps 40 --warmup 5 --pool 4 --no-cpu-baseline --no-scan-leg > $O/st_$st.json 2> $O/st_$st.err
  python - $O/st_$st.json $st <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
    print(f"stagger {sys.argv[2]:>4}  value {j['value']:8.1f} GB/s  ms/step {j['ms_per_step']:.4f}  {r['kernel']} {r['kernel_avg_us']:.1f} us")
except Exception as e:
    print("FAILED", e, open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
done
for st in 0 48; do ETLG_PLAN_STAGGER=$st timeout 120 python tools/plan_phases.py 2>&1 | tail -1; done
