"""ASYNC chain without a sidecar (frame_offsets = NULL), wall clock, per knob set: python tools/nosidecar_probe.py chained:ETLG_SCAN_CHAIN=1 host_count:ETLG_SCAN_CHAIN=0
(64 MiB cfg2 batches, device-resident, NO_CONTROL | ASYNC, `depth` in flight, every batch synced and its frame count checked)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
variants = sys.argv[1:] or ["chained:ETLG_SCAN_CHAIN=1", "host_count:ETLG_SCAN_CHAIN=0"]
w = synth.cfg2()
pool = []
for k in range(6):
    buf, offs = w.fill(64 << 20)
    pool.append((torch.from_numpy(buf.copy()).cuda(), len(buf), len(offs) - 1))
torch.cuda.synchronize()
FL = abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL | abi.F_ASYNC
N = int(os.environ.get("PROBE_N", "120"))
for v in variants:
    name, _, envs = v.partition(":")
    for k in ("ETLG_OVERLAP", "ETLG_SCAN_CHAIN", "PROBE_DEPTH"):
        os.environ.pop(k, None)
    for kv in filter(None, envs.split(",")):
        a, b = kv.split("=")
        os.environ[a] = b
    depth = int(os.environ.get("PROBE_DEPTH", "8"))
    d = Decoder(0)
    synth.cfg2().register(d)
    def run(n):
        infl = []
        for k in range(n):
            if len(infl) >= depth:
                b, nf = infl.pop(0); assert b.sync() == 0 and b.view().n_frames == nf, (b.error,); b.close()
            tb, nb, nf = pool[k % 6]
            infl.append((d.decode_device(tb.data_ptr(), nb, None, 0, FL), nf))
        for b, nf in infl:
            assert b.sync() == 0 and b.view().n_frames == nf; b.close()
    run(24)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); run(N); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / N)
    row = {"variant": name, "env": envs, "us_per_batch": round(best * 1e6, 1), "GBps": round((64 << 20) / best / 1e9, 1), "paths": d.debug_paths(), "scan_chained": d.debug_scan_chained(), "overlapped": d.debug_overlapped()}
    print(json.dumps(row), flush=True)
    d.close()
