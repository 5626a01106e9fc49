( time timeout 1500 python -m pytest tests -m gpu -q -x ) > gpurun_out/r06zr_tests.log 2>&1; tail -n 4 gpurun_out/r06zr_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
