O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/r06zc_tests.log 2>&1; tail -n 4 $O/r06zc_tests.log
( time timeout 900 python bench.py > $O/r06zc_bench.json 2> $O/r06zc_bench.err ); tail -c 200 $O/r06zc_bench.json
