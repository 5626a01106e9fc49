# wall-clock timeline of every k_plan3 tile (ETLG_PLAN_DBG=64), 64 MiB cfg2 batches back to back: python tools/plan3_timeline.py
import os, sys, ctypes
sys.path.insert(0, os.getcwd())
os.environ["ETLG_PLAN_DBG"] = str(64 | int(os.environ.get("DBG_EXTRA", "0")))
import numpy as np, torch
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
w = synth.cfg2(); d = Decoder(0); w.register(d)
pool = []
for k in range(6):
    buf, offs = w.fill(64 << 20)
    pool.append((torch.from_numpy(buf).cuda(), torch.from_numpy(offs.view(np.int32)).cuda(), len(buf), len(offs) - 1, offs))
torch.cuda.synchronize()
hip = ctypes.CDLL("libamdhip64.so")
for rep in range(2):
    keep = []
    for k in range(8):
        tb, to, nb, nfr, offs = pool[k % 6]
        keep.append(d.decode_device(tb.data_ptr(), nb, to.data_ptr(), nfr, abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL | abi.F_ASYNC))
    for x in keep: x.sync()
    v = keep[-1].view()
    nt = (len(offs) - 1 + 63) // 64
    host = torch.empty(nt * 8, dtype=torch.int64)
    hip.hipMemcpy(ctypes.c_void_p(host.data_ptr()), ctypes.c_void_p(ctypes.cast(v.heap, ctypes.c_void_p).value), nt * 64, 2)
    t = host.numpy().reshape(nt, 8).astype(np.float64)
    if rep == 0: continue
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0   # 100 MHz
    # slots: 0 iteration start | 6 next tile's LDS-DMA issued | 2 heads done (published) | 3 rows in registers | [next iteration of the wave:] 1 vmcnt(0) | 4 resolved | 5 stored
    order = [0, 6, 2, 3, 1, 4, 5]
    names = ["iter start", "DMA(i+1) issued", "heads+publish", "rows done", "vmcnt(0) [next iter]", "resolved", "stored"]
    print("tiles", nt, "kernel span %.1f us" % us[:, 5].max())
    for a, b in zip(range(len(order) - 1), range(1, len(order))):
        dlt = us[:, order[b]] - us[:, order[a]]
        q = np.percentile(dlt, [10, 50, 90, 100])
        print(f"  {names[a]:22s} -> {names[b]:22s}: p10 {q[0]:5.2f} p50 {q[1]:5.2f} p90 {q[2]:5.2f} max {q[3]:5.2f}")
    for lo in list(range(0, nt, max(1, nt // 12))):
        print("tile %5d: " % lo + "  ".join("%6.1f" % us[lo, k] for k in order))
    np.save("gpurun_out/plan3_timeline.npy", us)
