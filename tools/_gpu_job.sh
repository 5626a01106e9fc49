cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/finish_probe.py 2>&1 | tail -2 > gpurun_out/r06zq_finish_probe.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06z_prof -o p -- python tools/finish_probe.py > /dev/null 2>&1
f=$(ls gpurun_out/r06z_prof/*kernel_stats.csv | head -1); head -4 $f | cut -c1-160 >> gpurun_out/r06zq_finish_probe.txt
rm -rf gpurun_out/r06z_prof
timeout 900 python -m pytest tests/test_gpu_finish.py tests/test_gpu_columns.py tests/test_gpu_rowbinary.py tests/test_gpu_protobuf.py -q -x 2>&1 | tail -2 >> gpurun_out/r06zq_finish_probe.txt
python bench.py --legs=wide70,handoff --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('wide70', r['wide70']['value'], 'finish', r['wide70']['finish_cells']['value']); print({k:v['ms'] for k,v in r['handoff']['cfg3'].items() if isinstance(v,dict) and 'ms' in v})" >> gpurun_out/r06zq_finish_probe.txt 2>&1
