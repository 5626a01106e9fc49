# host-side enqueue cost per step (no kernel wait): python tools/hostcost.py
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
w = synth.cfg2(); d = Decoder(0); w.register(d)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); d.set_stream(stream.cuda_stream)
bufs = []
for _ in range(4):
    buf, offs = w.fill(64 << 20)
    bufs.append((torch.from_numpy(buf).cuda(), torch.from_numpy(offs.view(np.int32)).cuda(), len(buf), len(offs) - 1))
hdr = torch.zeros(8, dtype=torch.int64, device="cuda")
hdrs = torch.zeros((128, 8), dtype=torch.int64, device="cuda")
fl = abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL | abi.F_ASYNC
torch.cuda.synchronize()
def run(n, mode):
    keep = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        tb, to, nb, nf = bufs[k % 4]
        b = d.decode_device(tb.data_ptr(), nb, to.data_ptr(), nf, fl)
        if mode == "fixed": b.header_to_device(hdr.data_ptr())
        elif mode == "slots": b.header_to_device(hdrs[k].data_ptr())
        elif mode == "slots_ptr": b.header_to_device(hdrs.data_ptr() + 64 * k)
        keep.append(b)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    for b in keep: b.sync(); b.close()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
for mode in ("fixed", "slots", "slots_ptr", "none", "fixed"):
    run(20, mode)
    print(mode, "enqueue us/step %.1f  total us/step %.1f" % run(100, mode))
