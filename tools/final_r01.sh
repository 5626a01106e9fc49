# End-of-round measurement set: parity, bench lines, rocprofv3 kernel stats, HBM traffic passes.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest_gpu.log 2>&1; echo "parity rc=$?"; tail -2 gpurun_out/final/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 400 python bench.py > gpurun_out/final/bench_cfg2.json 2> gpurun_out/final/bench_cfg2.err; tail -1 gpurun_out/final/bench_cfg2.json | cut -c1-200
timeout 400 python bench.py --workload cfg3 > gpurun_out/final/bench_cfg3.json 2> gpurun_out/final/bench_cfg3.err; tail -1 gpurun_out/final/bench_cfg3.json | cut -c1-200
for wl in cfg2 cfg3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/stats_$wl -o p -- python bench.py --workload $wl --no-cpu-baseline > gpurun_out/final/stats_$wl.json 2> gpurun_out/final/stats_$wl.err
  ls gpurun_out/final/stats_$wl/ | head; cat gpurun_out/final/stats_$wl/*kernel_stats.csv 2>/dev/null | head -5
done
bash tools/traffic.sh cfg2
bash tools/traffic.sh cfg3
