python bench.py --legs=cfg2_mixed --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['value']); print(json.dumps(r['cfg2_mixed'], indent=0))" > gpurun_out/r06p_mixed.txt 2>&1
ETLG_CHAIN_REISSUE=0 python bench.py --legs=cfg2_mixed --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('reissue off', r['value']); print(json.dumps(r['cfg2_mixed'], indent=0))" >> gpurun_out/r06p_mixed.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_async.py tests/test_gpu_fixed_plan.py tests/test_gpu_copy.py tests/test_shim_twin.py -x -q 2>&1 | tail -3 >> gpurun_out/r06p_mixed.txt
