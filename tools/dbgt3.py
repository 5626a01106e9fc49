# per-phase shader-clock sums (ETLG_FUSED_DBG=8) for a 64 MiB batch: python tools/dbgt3.py [cfg2|cfg3]
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
os.environ["ETLG_FUSED_DBG"] = str(8 | int(os.environ.get("DBG_EXTRA", "0")))
import numpy as np, torch
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
w = getattr(synth, wl)(); d = Decoder(0); w.register(d)
bufs = [w.fill(64 << 20) for _ in range(1)]
for buf, offs in bufs:
    tb = torch.from_numpy(buf).cuda(); to = torch.from_numpy(offs.view(np.int32)).cuda(); torch.cuda.synchronize()
    b = d.decode_device(tb.data_ptr(), len(buf), to.data_ptr(), len(offs) - 1, abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL)
    out = (C.c_ulonglong * 12)(); d.L.etlg_ctx_debug_times(d.h, out)
    paths = d.debug_paths()
    blk = 256 if paths["fused"] and os.environ.get("ETLG_FUSED_KERNEL", "0") == "0" and wl == "cfg2" else 64
    nt = max(1, ((len(offs) - 1 + blk - 1) // blk) // 16)   # phase clocks are sampled: 1 tile in 16
    if paths["cells"]:
        names = ["offs", "stage", "P1 walk", "P2 heap size", "P2b sizes", "lookback", "positions", "P3 decode", "P4 finalize"]
    else:
        names = ["ticket+offs", "stage", "structure", "txn scans", "lookback1", "size", "out scans", "lookback2", "write"]
    print(paths, "tiles", nt, {n: round(out[i] / nt) for i, n in enumerate(names)}, "sum", round(sum(out[:12]) / nt), "extra", [round(out[i] / nt) for i in (9, 10, 11)])
