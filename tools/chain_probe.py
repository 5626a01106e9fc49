"""ASYNC chain rate of one context, wall clock, per workload and knob set: python tools/chain_probe.py cfg3 two_streams: one_stream:ETLG_OVERLAP=0
(64 MiB device-resident batches, sidecar, NO_CONTROL | ASYNC, 24 in flight, every batch synced and its frame count checked)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
wl = sys.argv[1]
variants = sys.argv[2:] or ["two_streams:", "one_stream:ETLG_OVERLAP=0"]
MK = {"cfg2": synth.cfg2, "cfg3": synth.cfg3}[wl]
w = MK()
pool = []
for k in range(6):
    buf, offs = w.fill(64 << 20)
    pool.append((torch.from_numpy(buf.copy()).cuda(), torch.from_numpy(offs.astype(np.uint32).view(np.int32).copy()).cuda(), len(buf), len(offs) - 1))
torch.cuda.synchronize()
FL = abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL | abi.F_ASYNC
rows = []
for v in variants:
    name, _, envs = v.partition(":")
    for k in ("ETLG_OVERLAP", "ETLG_FUSED_KERNEL", "ETLG_PLAN", "ETLG_ROWS"):
        os.environ.pop(k, None)
    for kv in filter(None, envs.split(",")):
        a, b = kv.split("=")
        os.environ[a] = b
    d = Decoder(0)
    MK().register(d)
    def run(n):
        infl = []
        for k in range(n):
            if len(infl) >= 24:
                b, nf = infl.pop(0); assert b.sync() == 0 and b.view().n_frames == nf, (b.error,); b.close()
            tb, to, nb, nf = pool[k % 6]
            infl.append((d.decode_device(tb.data_ptr(), nb, to.data_ptr(), nf, FL), nf))
        for b, nf in infl:
            assert b.sync() == 0 and b.view().n_frames == nf; b.close()
    run(40)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); run(120); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 120)
    row = {"workload": wl, "variant": name, "env": envs, "us_per_batch": round(best * 1e6, 1), "GBps": round((64 << 20) / best / 1e9, 1), "paths": {**d.debug_paths(), **d.debug_rows()}, "overlapped": d.debug_overlapped()}
    print(json.dumps(row), flush=True)
    rows.append(row)
    d.close()
