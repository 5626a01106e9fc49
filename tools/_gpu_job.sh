( time timeout 1500 python -m pytest tests -m gpu -q -x ) > gpurun_out/r06zg_tests.log 2>&1; tail -n 4 gpurun_out/r06zg_tests.log
