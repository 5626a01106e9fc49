timeout 400 python tools/finish_fuzz.py 240 1000 2>&1 | tail -3 > gpurun_out/r06zp_finish_fuzz.txt
timeout 300 python tools/async_fuzz.py 2>&1 | tail -1 >> gpurun_out/r06zp_finish_fuzz.txt
timeout 300 python tools/scan_hunt.py 2>&1 | tail -2 >> gpurun_out/r06zp_finish_fuzz.txt
