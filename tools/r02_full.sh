#!/bin/bash
# full GPU suite + default bench line
O=gpurun_out/r02f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json; tail -3 $O/bench.err
