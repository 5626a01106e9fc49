python tools/wide_ab.py rows > gpurun_out/r06l_wide.txt 2>&1
python tools/rows_ab.py rows > gpurun_out/r06_ab.txt 2>&1
ETLG_ROWS=0 python tools/rows_ab.py cells >> gpurun_out/r06_ab.txt 2>&1
ETLG_FUSED_DBG=8 python tools/rows_ab.py rows_phases >> gpurun_out/r06_ab.txt 2>&1
timeout 600 python tools/chain_probe.py cfg3 rows_two: 2>&1 | grep workload >> gpurun_out/r06l_wide.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4 > gpurun_out/r06l_tests.txt
