import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
mode = sys.argv[1:]
w = synth.cfg2()
pool = []
mib = 64
for a in mode:
    if a.startswith("mib="): mib = int(a[4:])
for k in range(6):
    buf, offs = w.fill(mib << 20)
    pool.append((torch.from_numpy(buf.copy()).cuda(), torch.from_numpy(offs.astype(np.uint32).view(np.int32).copy()).cuda(), len(buf), len(offs) - 1))
torch.cuda.synchronize()
d = Decoder(0)
if "userstream" in mode:
    st = torch.cuda.Stream(); torch.cuda.set_stream(st); d.set_stream(st.cuda_stream)
synth.cfg2().register(d)
hdrs = torch.zeros((4096, 8), dtype=torch.int64, device="cuda")
FL = abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL | abi.F_ASYNC
infl = []
bad = []
n = 0
def retire():
    global n
    b, nf, k = infl.pop(0)
    rc = b.sync(); v = b.view()
    if rc != 0 or v.n_frames != nf: bad.append((k, rc, int(v.n_events), int(v.n_frames), nf))
    b.close()
for k in range(600):
    tb, to, nb, nf = pool[k % 6]
    if len(infl) >= 24: retire()
    b = d.decode_device(tb.data_ptr(), nb, to.data_ptr(), nf, FL)
    if "hdr" in mode: b.header_to_device(hdrs[k % 4096].data_ptr())
    infl.append((b, nf, k))
while infl: retire()
print(json.dumps({"mode": mode, "bad": bad[:12], "nbad": len(bad), "paths": d.debug_paths(), "overlapped": d.debug_overlapped()}))
