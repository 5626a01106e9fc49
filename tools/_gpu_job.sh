# same-box comparison: k_rows (default), k_cells (ETLG_ROWS=0), optionally a previous build (etl_amd/_prev.so)
python tools/rows_ab.py rows > gpurun_out/r06_ab.txt 2>&1
ETLG_ROWS=0 python tools/rows_ab.py cells >> gpurun_out/r06_ab.txt 2>&1
[ -f etl_amd/_prev.so ] && ETLG_LIB_PATH=etl_amd/_prev.so python tools/rows_ab.py prev >> gpurun_out/r06_ab.txt 2>&1
ETLG_FUSED_DBG=8 python tools/rows_ab.py rows_phases >> gpurun_out/r06_ab.txt 2>&1
python tools/rows_ab.py rows_again >> gpurun_out/r06_ab.txt 2>&1
