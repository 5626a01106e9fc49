"""k_cells on cfg3: kernel time (library HIP events) and per-phase clocks, for the shipped kernel and for the
profiling ablations of the P3 value codecs (ETLG_FUSED_DBG bits 6..10: numeric, date/time, uuid, text copy, others).
Ablated runs produce wrong results by design; they price a family of codecs, nothing else.
usage: python tools/cells_probe.py [ablation masks ...]   (default: 0 1 2 4 8 16 31)"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import abi, synth
from etl_amd.decoder import Decoder

NAMES = {0: "P0 stage", 1: "vote", 11: "P1 classify", 9: "P1 walk", 10: "P1 txn scan", 2: "P1 slot+barrier", 3: "P2 sizing", 4: "P2b prefix/scan",
         5: "look-back", 6: "ctx distribute", 7: "P3 decode", 8: "P4 headers"}
masks = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 8, 16, 31]
w = synth.cfg3()
buf, offs = w.fill(64 << 20)
tb = torch.from_numpy(buf.copy()).cuda()
to = torch.from_numpy(offs.astype(np.uint32).view(np.int32).copy()).cuda()
nf = len(offs) - 1
out_rows = []
for phases in (False, True):
    for m in masks:
        if not phases and m:
            continue
        dbg = (8 if phases else 0) | (m << 6)
        os.environ["ETLG_FUSED_DBG"] = str(dbg)
        d = Decoder(0)
        w2 = synth.cfg3()
        w2.register(d)
        d.profile(True)
        t = None
        n0 = ms0 = 0
        for it in range(4 + 16):
            if it == 4:  # the first launches of a context are slower (code load, arena growth): not timed
                n0, ms0 = d.profile_read().get("k_cells", (0, 0.0))
            b = d.decode_device(tb.data_ptr(), tb.numel(), to.data_ptr(), nf, abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL)
            if phases:
                o = (C.c_ulonglong * 12)()
                d.L.etlg_ctx_debug_times(d.h, o)
                t = [int(x) for x in o]
            b.close()
        prof = d.profile_read()
        n, ms = prof.get("k_cells", (0, 0.0))
        n, ms = n - n0, ms - ms0
        row = {"dbg": dbg, "ablate": m, "k_cells_us": round(1e3 * ms / max(n, 1), 1), "launches": n, "paths": d.debug_paths()}
        if t:
            tot = sum(t)
            ntile = max(((nf + 63) // 64) // 16, 1)
            row["cycles_per_tile"] = tot // ntile
            row["phases_pct"] = {NAMES[k]: round(100.0 * t[k] / tot, 1) for k in NAMES}
        print(json.dumps(row), flush=True)
        out_rows.append(row)
        d.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out_rows, open("gpurun_out/cells_probe.json", "w"), indent=1)
