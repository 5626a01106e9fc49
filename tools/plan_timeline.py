# wall-clock timeline of every k_plan tile (ETLG_PLAN_DBG=64) for one 64 MiB cfg2 batch: python tools/plan_timeline.py
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
os.environ["ETLG_PLAN_DBG"] = str(64 | int(os.environ.get("DBG_EXTRA", "0")))
import numpy as np, torch
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
w = synth.cfg2(); d = Decoder(0); w.register(d)
pool = []
for k in range(6):   # bench.py conditions: a pool larger than the Infinity Cache, batches decoded back to back
    buf, offs = w.fill(64 << 20)
    pool.append((torch.from_numpy(buf).cuda(), torch.from_numpy(offs.view(np.int32)).cuda(), len(buf), len(offs) - 1, offs))
torch.cuda.synchronize()
for rep in range(2):
    keep = []
    for k in range(8):
        tb, to, nb, nfr, offs = pool[k % 6]
        keep.append(d.decode_device(tb.data_ptr(), nb, to.data_ptr(), nfr, abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL | abi.F_ASYNC))
    for x in keep: x.sync()
    b = keep[-1]
    v = b.view()
    nt = (len(offs) - 1 + 63) // 64
    raw = (C.c_ulonglong * (nt * 8)).from_address(0)
    host = torch.empty(nt * 8, dtype=torch.int64)
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy(ctypes.c_void_p(host.data_ptr()), ctypes.c_void_p(ctypes.cast(v.heap, ctypes.c_void_p).value), nt * 64, 2)
    t = host.numpy().reshape(nt, 8).astype(np.float64)
    t0 = t[:, 0].min()
    us = (t[:, :6] - t0) / 100.0   # 100 MHz
    names = ["start", "data landed", "heads done (publish)", "look-back done", "txn ctx done", "rows written"]
    if rep == 1:
        print("tiles", nt, "kernel span %.1f us" % us[:, 5].max())
        for k, n in enumerate(names):
            q = np.percentile(us[:, k], [0, 10, 50, 90, 100])
            print(f"{n:24s} min {q[0]:6.1f}  p10 {q[1]:6.1f}  p50 {q[2]:6.1f}  p90 {q[3]:6.1f}  max {q[4]:6.1f}")
        for k in range(1, 6):
            dlt = us[:, k] - us[:, k - 1]
            q = np.percentile(dlt, [10, 50, 90, 100])
            print(f"  {names[k-1]} -> {names[k]}: p10 {q[0]:5.1f} p50 {q[1]:5.1f} p90 {q[2]:5.1f} max {q[3]:5.1f}")
        for lo in range(0, nt, max(1, nt // 16)):
            print("tile %5d: " % lo + "  ".join("%6.1f" % x for x in us[lo, :6]))
        np.save("gpurun_out/plan_timeline.npy", us)
