mkdir -p gpurun_out/r06g
( time timeout 900 python bench.py > gpurun_out/r06g/bench3.json 2> gpurun_out/r06g/bench3.err ); tail -c 100 gpurun_out/r06g/bench3.json
