cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/parity_all.log 2>&1; echo "parity rc=$?" 
tail -3 gpurun_out/parity_all.log
for nb in 0 1; do ETLG_NO_ROW_BUFFER_X=$nb; if [ $nb = 1 ]; then export ETLG_NO_ROW_BUFFER=1; fi; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nobuf=$nb', d['value'], d['roofline']['kernel_avg_us'], d['roofline']['frac'], d['roofline']['alg_bytes_per_launch'])"; done
