O=gpurun_out/r06zm_tiles.txt
ETLG_ROWS_TRACE=1 python tools/wide_ab.py auto 2>&1 | grep -E "^\{|planned" | sort -u | cut -c1-400 > $O
ETLG_ROWS_TRACE=1 ETLG_ROWS_CF=24 python tools/wide_ab.py cf24 2>&1 | grep -E "^\{|planned" | sort -u | cut -c1-400 >> $O
python tools/rows_ab.py auto 2>&1 | grep -E "^\{" | cut -c1-700 >> $O
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > gpurun_out/r06zm_tests.log 2>&1; tail -n 4 gpurun_out/r06zm_tests.log
