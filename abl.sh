for n in w3s2 w4s2 w3s16 w4s16; do
  echo "== $n"; ETLG_LIB_PATH=$PWD/etl_amd/variants/lib_$n.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/abl_$n.log 2>&1
  grep metric gpurun_out/abl_$n.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['pipeline_kernels_us'])" || tail -5 gpurun_out/abl_$n.log
done
