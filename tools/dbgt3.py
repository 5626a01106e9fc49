import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
os.environ["ETLG_FUSED_DBG"] = "8"
import numpy as np, torch
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
w = synth.cfg3(); d = Decoder(0); w.register(d)
bufs = [w.fill(64 << 20) for _ in range(2)]
for buf, offs in bufs:
    tb = torch.from_numpy(buf).cuda(); to = torch.from_numpy(offs.view(np.int32)).cuda(); torch.cuda.synchronize()
    b = d.decode_device(tb.data_ptr(), len(buf), to.data_ptr(), len(offs) - 1, abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL)
    out = (C.c_ulonglong * 12)(); d.L.etlg_ctx_debug_times(d.h, out)
    blk = 64
    nt = (len(offs) - 1 + blk - 1) // blk
    names = ["ticket+offs", "stage", "structure", "txn scans", "lookback1", "size", "out scans", "lookback2", "write"]
    print("tiles", nt, {n: round(out[i] / nt) for i, n in enumerate(names)}, "sum", round(sum(out[:9]) / nt))
