"""Do two decode kernels of DIFFERENT batches overlap usefully on one MI355X?  N contexts (one stream each) decode cfg2 batches
concurrently, issued round robin from one host thread (ASYNC, device-resident); aggregate WAL GB/s against one context alone.
usage: python tools/plan_dual.py [name:ENV=val,... ...]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import abi, synth
from etl_amd.decoder import Decoder

variants = [a for a in sys.argv[1:] if not a.startswith("wl=")] or ["default:"]
WL = ([a[3:] for a in sys.argv[1:] if a.startswith("wl=")] or ["cfg2"])[0]
MK = {"cfg2": synth.cfg2, "cfg3": synth.cfg3}[WL]
KNOBS = ("ETLG_PLAN_DBG", "ETLG_OVERLAP", "ETLG_PLAN_MARGIN")
NCTX = 3
ws = [MK(seed=0xE710002 + 97 * k) for k in range(NCTX)]
pools = []
for w in ws:
    pool = []
    for k in range(3):
        buf, offs = w.fill(64 << 20)
        pool.append((torch.from_numpy(buf.copy()).cuda(), torch.from_numpy(offs.astype(np.uint32).view(np.int32).copy()).cuda(), len(buf), len(offs) - 1))
    pools.append(pool)
torch.cuda.synchronize()
FL = abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL | abi.F_ASYNC
rows = []
for v in variants:
    name, _, envs = v.partition(":")
    for k in KNOBS:
        os.environ.pop(k, None)
    for kv in filter(None, envs.split(",")):
        a, b = kv.split("=")
        os.environ[a] = b
    for nctx in (1, 2, 3):
        decs = []
        for k in range(nctx):
            d = Decoder(0)
            ws[k].register(d)
            decs.append(d)
        def run(rounds):
            keep = []
            for r in range(rounds):
                for k, d in enumerate(decs):
                    tb, to, nb, nf = pools[k][r % 3]
                    keep.append(d.decode_device(tb.data_ptr(), nb, to.data_ptr(), nf, FL))
            nbytes = 0
            for b in keep:
                b.sync()
                assert b.rc == 0, b.error
                nbytes += 64 << 20
                b.close()
            return nbytes
        run(6); run(6)
        torch.cuda.synchronize()
        best = 0.0
        for rep in range(3):
            t0 = time.perf_counter()
            nb = 0
            for _ in range(4):
                nb += run(6)
            dt = time.perf_counter() - t0
            best = max(best, nb / dt / 1e9)
        row = {"workload": WL, "variant": name, "contexts": nctx, "GBps": round(best, 1), "us_per_batch": round((64 << 20) / best / 1e3, 1), "paths": decs[0].debug_paths()}
        print(json.dumps(row), flush=True)
        rows.append(row)
        for d in decs:
            d.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/plan_dual.json", "w"), indent=1)
