# PCIe-inclusive decode rate (host buffers in, optionally host arena out): python tools/pcie_rate.py
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
for wl in ("cfg2", "cfg3"):
    w = getattr(synth, wl)(); d = Decoder(0); w.register(d)
    bufs = [w.fill(64 << 20) for _ in range(3)]
    pinned = [(torch.from_numpy(b).pin_memory().numpy(), o) for b, o in bufs]
    for name, src, flags in (("pageable in, host out", bufs, abi.F_NO_CONTROL), ("pageable in, device out", bufs, abi.F_NO_CONTROL | abi.F_OUTPUT_ON_DEVICE),
                             ("pinned in, device out", pinned, abi.F_NO_CONTROL | abi.F_OUTPUT_ON_DEVICE), ("pinned in, device out, no sidecar", [(b, None) for b, _ in pinned], abi.F_NO_CONTROL | abi.F_OUTPUT_ON_DEVICE)):
        for b, o in src[:2]:
            d.decode(b, o, flags=flags).close()
        torch.cuda.synchronize(); t0 = time.perf_counter(); nb = 0
        for k in range(6):
            b, o = src[k % 3]
            r = d.decode(b, o, flags=flags); assert r.rc == 0; nb += len(b); r.close()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(wl, name, "%.1f GB/s" % (nb / (t1 - t0) / 1e9))
    d.close()
