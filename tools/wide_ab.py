"""The reference's 68-column type-matrix table through the default kernel choice: kernel time (HIP events) and paths.
usage: python tools/wide_ab.py [label]   (measurement tool, not product)"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
label = sys.argv[1] if len(sys.argv) > 1 else "default"
buf, offs = synth.type_matrix_stream(24000, mix=True)
tb = torch.from_numpy(buf.copy()).cuda(); to = torch.from_numpy(offs.astype(np.uint32).view(np.int32).copy()).cuda()
d = Decoder(0)
synth.type_matrix_register(d)
d.profile(True)
base = {}
for it in range(3 + 8):
    if it == 3:
        base = d.profile_read()
    b = d.decode_device(tb.data_ptr(), tb.numel(), to.data_ptr(), len(offs) - 1, abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL)
    assert b.rc == 0, b.error
    b.close()
prof = d.profile_read()
out = {}
for k, (n, ms) in prof.items():
    n0, ms0 = base.get(k, (0, 0.0))
    if n > n0:
        out[k] = round(1e3 * (ms - ms0) / (n - n0), 1)
us = sum(out.values())
print(json.dumps({"label": label, "bytes": int(len(buf)), "frames": int(len(offs) - 1), "kernels_us": out, "GBps_kernel": round(len(buf) / us / 1e3, 1), "paths": {**d.debug_paths(), **d.debug_rows()}}), flush=True)
d.close()
