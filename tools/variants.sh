# Times the flag variants built by tools/build_variants.py back to back (kernel average from the library's HIP events).
# usage (GPU box): bash tools/variants.sh "base nounroll os maxilp waves5 noinl" "cfg2 cfg3"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/variants
for wl in ${2:-cfg2 cfg3}; do
  for v in ${1:-base}; do
    ETLG_LIB_PATH=$GRAFT_REPO_ROOT/etl_amd/variants/libetl_gfx950_$v.so timeout 120 python bench.py --workload $wl --steps 30 --warmup 3 --pool 4 \
      --no-cpu-baseline --no-scan-leg > gpurun_out/variants/${v}_$wl.json 2> gpurun_out/variants/${v}_$wl.err
    python - $v $wl <<'PY'
import json, sys
v, wl = sys.argv[1:3]
try:
    j = json.loads(open(f"gpurun_out/variants/{v}_{wl}.json").read().strip().splitlines()[-1])
    r = j["roofline"]
    print(f"{v:9s} {wl} value {j['value']:8.1f} GB/s  ms/step {j['ms_per_step']:.4f}  {r['kernel']} {r['kernel_avg_us']:.1f} us  frac {r['frac']}")
except Exception as e:
    print(v, wl, "FAILED", e)
PY
  done
done
