python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/abl_x.log 2>&1
grep metric gpurun_out/abl_x.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['pipeline_kernels_us'])" || tail -5 gpurun_out/abl_x.log
