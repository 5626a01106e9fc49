cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export ETLG_LIB_PATH=$PWD/build/variants/noexact.so
python tools/finish_probe.py 2>&1 | tail -1 > gpurun_out/r06zs_noexact.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06z_prof -o p -- python tools/finish_probe.py > /dev/null 2>&1
f=$(ls gpurun_out/r06z_prof/*kernel_stats.csv | head -1); head -3 $f | cut -c1-160 >> gpurun_out/r06zs_noexact.txt
rm -rf gpurun_out/r06z_prof
