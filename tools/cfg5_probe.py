"""cfg5 leg of bench.py alone (control path, default flags + ASYNC), with and without the pre-pass running ahead:
python tools/cfg5_probe.py [npool] [passes]   — ETLG_HOST_TIMES=1 prints the host's wall-clock split when the context goes."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
npool = int(sys.argv[1]) if len(sys.argv) > 1 else 8
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
for env in ({}, {"ETLG_CTL_ASYNC": "0"}):
    for k in ("ETLG_CTL_ASYNC",):
        os.environ.pop(k, None)
    os.environ.update(env)
    r = bench.leg_cfg5(0, dev, 64 << 20, npool, passes)
    r["env"] = env
    print(json.dumps(r), flush=True)
