"""Phase clocks of k_cells (ETLG_FUSED_DBG=8): share of a sampled tile's time per phase (the spine wave's first lane, every 16th tile).
usage: python tools/cells_phases.py [ablation bits] [wl=cfg3|cfg5]   (cfg5: default flags, the stream's second batch)"""
import ctypes as C
import os
import sys

ARGS = [a for a in sys.argv[1:] if not a.startswith("wl=")]
WL = ([a[3:] for a in sys.argv[1:] if a.startswith("wl=")] or ["cfg3"])[0]
os.environ["ETLG_FUSED_DBG"] = str(8 | (int(ARGS[0]) << 6)) if ARGS else "8"
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import abi, synth
from etl_amd.decoder import Decoder

w = getattr(synth, WL)()
d = Decoder(0)
w.register(d, ready=(WL != "cfg5"))
buf, offs = w.fill(64 << 20)
FL = abi.F_OUTPUT_ON_DEVICE | (abi.F_NO_CONTROL if WL != "cfg5" else 0)
tb = torch.from_numpy(buf.copy()).cuda()
to = torch.from_numpy(offs.astype(np.uint32).view(np.int32).copy()).cuda()
names = {0: "P0 stage+side+offsets", 1: "window vote", 9: "P1 walk", 10: "P1 txn scan(+seq lookback)", 2: "P1 ownership/slot + barrier",
         3: "P2 heap sizing", 4: "P2b shapes/prefix/scan", 5: "look-back", 6: "ctx/prefix distribution", 7: "P3 decode+write", 8: "P4 headers/states"}
for it in range(3):
    if WL == "cfg5" and it:   # a stream with DDL: consecutive batches, not the same one again
        buf, offs = w.fill(64 << 20)
        tb = torch.from_numpy(buf.copy()).cuda()
        to = torch.from_numpy(offs.astype(np.uint32).view(np.int32).copy()).cuda()
    b = d.decode_device(tb.data_ptr(), tb.numel(), to.data_ptr(), len(offs) - 1, FL)
    assert b.rc == 0, b.error
    out = (C.c_ulonglong * 12)()
    d.L.etlg_ctx_debug_times(d.h, out)
    t = [int(x) for x in out]
    b.close()
tot = sum(t)
nt = (len(offs) - 1 + 63) // 64 // 16
print("sampled tiles ~", nt, "cycles/tile", tot // max(nt, 1))
for k in (0, 1, 9, 10, 2, 3, 4, 5, 6, 7, 8):
    print(f"{names[k]:34s} {100.0 * t[k] / tot:5.1f} %  {t[k] // max(nt, 1):7d} cyc")
print(d.debug_paths())
