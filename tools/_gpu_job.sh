O=gpurun_out/r06zo_persist.txt
for w in 0 8 6 4 12; do ETLG_PLAN_PERSIST=$w python bench.py --legs= --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('persist $w', r['value'], r['roofline']['kernel_avg_us'], r['roofline']['alone_us'], r['roofline'].get('pipeline_kernels_us'))" >> $O 2>&1; done
ETLG_PLAN_PERSIST=8 timeout 900 python -m pytest tests/test_gpu_fixed_plan.py tests/test_gpu_async.py -x -q 2>&1 | tail -3 >> $O
