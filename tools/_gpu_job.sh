timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06k_gpu_tests.txt
python tools/rows_ab.py rows > gpurun_out/r06_ab.txt 2>&1
ETLG_ROWS=0 python tools/rows_ab.py cells >> gpurun_out/r06_ab.txt 2>&1
