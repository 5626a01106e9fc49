# full GPU suite, the default bench line (all legs), and a small cfg4 run
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -3 $O/pytest.log | tr '\n' ' ')"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - $O/bench.json <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(f"cfg2 value {j['value']} GB/s read_frac {j['hbm_read_frac']} ms/step {j['ms_per_step']} {r['kernel']} {r['kernel_avg_us']} us frac {r['frac']} paths {j['config']['kernel_paths']}")
    for k in ("no_sidecar", "cfg3", "cfg5", "copy"):
        if k in j:
            x = j[k]; print(k, x["value"], "GB/s", x.get("roofline", {}).get("pipeline_kernels_us") or x.get("kernels_us"), x.get("paths"))
    print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"].get("all_cores", {}).get("value"))
except Exception as e:
    print("FAILED", e)
PY
timeout 600 python bench.py --workload cfg4 --cfg4-gib 2 --cfg4-seg-mib 256 > $O/cfg4_small.json 2> $O/cfg4_small.err; echo "cfg4 rc=$?"; tail -2 $O/cfg4_small.err; head -c 1500 $O/cfg4_small.json
