cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/finish_probe.py 2>&1 | tail -2 > gpurun_out/r06za_finish_probe.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06z_prof -o p -- python tools/finish_probe.py > /dev/null 2>&1
f=$(ls gpurun_out/r06z_prof/*kernel_stats.csv | head -1); head -4 $f | cut -c1-160 >> gpurun_out/r06za_finish_probe.txt
rm -rf gpurun_out/r06z_prof
timeout 900 python -m pytest tests/test_gpu_finish.py -q -x 2>&1 | tail -2 >> gpurun_out/r06za_finish_probe.txt
