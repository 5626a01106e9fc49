"""The boundary scan alone on one device-resident 64 MiB cfg2 / cfg3 batch: wall time per call of etlg_scan_boundaries (device in / out),
for `rocprofv3 --kernel-trace --stats -- python tools/scan_alone.py` (per-kernel durations of the three scan kernels)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import synth
from etl_amd.decoder import Decoder
for mk in (synth.cfg2, synth.cfg3):
    w = mk()
    buf, offs = w.fill(64 << 20)
    tb = torch.from_numpy(buf.copy()).cuda()
    to = torch.empty(len(buf) // 24 + 2048, dtype=torch.int32, device="cuda")
    d = Decoder(0)
    for _ in range(5):
        nf = d.scan_boundaries_device(tb.data_ptr(), tb.numel(), to.data_ptr(), to.numel())
    assert nf == len(offs) - 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 100
    for _ in range(N):
        d.scan_boundaries_device(tb.data_ptr(), tb.numel(), to.data_ptr(), to.numel())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    print(w.name, "scan call", round(dt * 1e6, 1), "us", round(len(buf) / dt / 1e9, 1), "GB/s", flush=True)
    d.close()
