O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/r06v_tests.log 2>&1; tail -n 4 $O/r06v_tests.log
python tools/nosidecar_probe.py chained:ETLG_SCAN_CHAIN=1 host_count:ETLG_SCAN_CHAIN=0 > $O/r06v_nosidecar_probe.txt 2>&1
for m in nosidecar mixed cfg2; do timeout 200 python tools/async_long_fuzz.py 45 6 $m 2>&1 | tail -2 >> $O/r06v_long_fuzz.txt; done
( time timeout 900 python bench.py > $O/r06v_bench.json 2> $O/r06v_bench.err ); tail -c 300 $O/r06v_bench.json
