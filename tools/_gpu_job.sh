for w in 24 12 6 3; do ETLG_BENCH_WINDOW=$w python bench.py --legs= --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('window $w', r['value'], r['roofline']['kernel_avg_us'], r['roofline']['alone_us'])" >> gpurun_out/r06ze_window.txt 2>&1; done
