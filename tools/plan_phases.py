# per-phase shader-clock sums of k_plan (ETLG_PLAN_DBG=32; one tile in 16 sampled, lane 0) for a 64 MiB cfg2 batch: python tools/plan_phases.py
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
os.environ["ETLG_PLAN_DBG"] = str(32 | int(os.environ.get("DBG_EXTRA", "0")))
import numpy as np, torch
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
w = synth.cfg2(); d = Decoder(0); w.register(d)
names = ["start+offsets", "LDS-DMA wait", "heads+table+publish", "rest of heads+rows", "Begin-LSN word", "txn ctx+row copy", "event header stores", "look-back resolve"]
for rep in range(3):
    buf, offs = w.fill(64 << 20)
    tb = torch.from_numpy(buf).cuda(); to = torch.from_numpy(offs.view(np.int32)).cuda(); torch.cuda.synchronize()
    b = d.decode_device(tb.data_ptr(), len(buf), to.data_ptr(), len(offs) - 1, abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL)
    out = (C.c_ulonglong * 12)(); d.L.etlg_ctx_debug_times(d.h, out)
    nt = max(1, ((len(offs) - 1 + 63) // 64) // 16)
    print(d.debug_paths(), "sampled tiles", nt, {n: round(out[i] / nt) for i, n in enumerate(names)}, "sum", round(sum(out[:8]) / nt), "cycles per tile")
