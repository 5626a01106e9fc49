"""Hand-off calls on one decoded 64 MiB batch (cfg3 by default), repeated: wall time per call and — under
`rocprofv3 --kernel-trace --stats` — the kernels behind them.  usage: python tools/handoff_probe.py [cfg2|cfg3] [reps]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import abi, synth
from etl_amd.decoder import Decoder

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
w = getattr(synth, wl)()
buf, offs = w.fill(64 << 20)
tb = torch.from_numpy(buf.copy()).cuda()
to = torch.from_numpy(offs.astype(np.uint32).view(np.int32).copy()).cuda()
d = Decoder(0)
w.register(d)
b = d.decode_device(tb.data_ptr(), tb.numel(), to.data_ptr(), len(offs) - 1, abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL)
assert b.rc == 0, b.error
c0 = b.columns(0, kinds=("I", "U"), on_device=True)
nc = c0.view.n_cols
flags = [1 if c0.column(i).nullable else 0 for i in range(nc)] + [0, 0]
c0.close()
out = {"workload": wl, "reps": reps}
for name, call in (("columns", lambda: b.columns(0, kinds=("I", "U"), on_device=True)),
                   ("rowbinary", lambda: b.rowbinary(0, flags, abi.CH_MERGE_TREE, on_device=True)),
                   ("protobuf", lambda: b.protobuf(0, on_device=True))):
    for _ in range(3):
        call().close()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for _ in range(reps):
        r = call()
        n = getattr(r, "n_rows", 0)
        r.close()
    torch.cuda.synchronize()
    out[name] = {"ms_per_call": round(1e3 * (time.perf_counter() - t0) / reps, 4), "rows": int(n)}
print(json.dumps(out))
b.close()
d.close()
