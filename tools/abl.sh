for cfg in "256 1" "256 0" "64 1" "64 0"; do set -- $cfg
echo "== blk $1 dbg $2"; ETLG_FUSED_BLK=$1 ETLG_FUSED_DBG=$2 python bench.py --workload cfg3 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/abl_x.log 2>&1
grep metric gpurun_out/abl_x.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['pipeline_kernels_us'])" || tail -5 gpurun_out/abl_x.log
done
