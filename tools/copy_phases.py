"""Where k_cells' time goes on table-copy rows: phase clocks (ETLG_FUSED_DBG=8) and value-codec ablations (bits 6..10: numeric,
temporal, uuid, string / deferred text, the rest) for the copy leg's two kinds of rows. python tools/copy_phases.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import synth
from etl_amd.decoder import Decoder

names = {0: "P0 stage+side+offsets", 1: "window vote", 11: "P1 classify / copy split", 9: "P1 walk", 10: "P1 txn scan(+seq lookback)", 2: "P1 ownership/slot + barrier (+ copy_fix)",
         3: "P2 heap sizing", 4: "P2b shapes/prefix/scan", 5: "look-back", 6: "ctx/prefix distribution", 7: "P3 decode+write", 8: "P4 headers/states"}
for clean in (True, False):
    rows = synth.copy_rows(20000, 1, clean=clean) * 20
    buf = np.frombuffer(b"".join(rows), dtype=np.uint8)
    offs = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint32)
    tb = torch.from_numpy(buf.copy()).cuda()
    to = torch.from_numpy(offs.view(np.int32).copy()).cuda()
    for dbg in (0, 8):
        os.environ["ETLG_FUSED_DBG"] = str(dbg)
        d = Decoder(0)
        d.schema_put(42, 0, synth.COPY_COLS)
        slot = d.table_ready(42, 0, [1] * 10, [1] + [0] * 9)
        for _ in range(2):
            d.copy_decode_device(slot, tb.data_ptr(), tb.numel(), to.data_ptr(), len(rows)).close()
        if dbg & 8:
            b = d.copy_decode_device(slot, tb.data_ptr(), tb.numel(), to.data_ptr(), len(rows))
            out = (C.c_ulonglong * 12)()
            d.L.etlg_ctx_debug_times(d.h, out)
            t = [int(x) for x in out]
            b.close()
            tot = sum(t)
            nt = (len(rows) + 63) // 64 // 16
            print("clean" if clean else "escape-heavy", "dbg", dbg, "sampled tiles ~", nt, "cycles/tile", tot // max(nt, 1))
            for k in (0, 1, 11, 9, 10, 2, 3, 4, 5, 6, 7, 8):
                print(f"  {names[k]:34s} {100.0 * t[k] / tot:5.1f} %  {t[k] // max(nt, 1):7d} cyc")
        else:
            d.profile(True)
            for _ in range(5):
                d.copy_decode_device(slot, tb.data_ptr(), tb.numel(), to.data_ptr(), len(rows)).close()
            torch.cuda.synchronize()
            prof = {k: round(1000 * ms / max(c, 1), 1) for k, (c, ms) in d.profile_read().items() if c}
            print("clean" if clean else "escape-heavy", "dbg", dbg, prof, d.debug_paths())
        d.close()
