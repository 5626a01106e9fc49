cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_scan.py -m gpu -x -q > gpurun_out/scan.log 2>&1; echo "scan rc=$?"; tail -5 gpurun_out/scan.log
bash tools/run_abl.sh
for wl in cfg2 cfg3; do timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['value'], d['roofline']['kernel_avg_us'], d['no_sidecar'])"; done
