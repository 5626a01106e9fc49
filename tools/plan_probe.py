"""k_plan variants on cfg2: per 64 MiB batch, kernel time (library HIP events on the stream of each launch — with two decode
streams consecutive kernels overlap, so their durations add up to more than the wall clock) and wall-clock time per batch of an
ASYNC chain, all in ONE process on one box (boxes differ by up to 10 %, so only numbers of the same call compare). Batches
rotate through a pool larger than the Infinity Cache.
usage: python tools/plan_probe.py [variant ...]     variant = name:ENV=val,ENV=val   (default: a standard ladder)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import abi, synth
from etl_amd.decoder import Decoder

DEFAULT = ["two_streams:", "one_stream:ETLG_OVERLAP=0", "one_tile_per_wave:ETLG_PLAN_DBG=512", "two_streams_again:"]
variants = sys.argv[1:] or DEFAULT
KNOBS = ("ETLG_PLAN_DBG", "ETLG_OVERLAP", "ETLG_PLAN_MARGIN", "ETLG_PLAN_PRE")
w = synth.cfg2()
pool = []
for k in range(6):
    buf, offs = w.fill(64 << 20)
    pool.append((torch.from_numpy(buf.copy()).cuda(), torch.from_numpy(offs.astype(np.uint32).view(np.int32).copy()).cuda(), len(buf), len(offs) - 1))
torch.cuda.synchronize()
rows = []
for v in variants:
    name, _, envs = v.partition(":")
    for k in KNOBS:
        os.environ.pop(k, None)
    for kv in filter(None, envs.split(",")):
        a, b = kv.split("=")
        os.environ[a] = b
    d = Decoder(0)
    synth.cfg2().register(d)
    PROF = os.environ.get("PROBE_PROF", "1") != "0"
    d.profile(PROF)
    base = {}
    t_enq = 0.0
    import time
    t0 = 0.0
    for it in range(6 + 24):
        if it == 6:
            base = d.profile_read() if PROF else {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        keep = []
        te = time.perf_counter()
        for k in range(6):
            tb, to, nb, nf = pool[k]
            keep.append(d.decode_device(tb.data_ptr(), nb, to.data_ptr(), nf, abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL | abi.F_ASYNC))
        if it >= 6:
            t_enq += time.perf_counter() - te
        for b in keep:
            b.sync()
            assert b.rc == 0, b.error
            b.close()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    prof = d.profile_read() if PROF else {}
    row = {"variant": name, "host_enqueue_us_per_batch": round(1e6 * t_enq / (24 * 6), 2), "env": envs, "paths": d.debug_paths(), "overlapped": d.debug_overlapped(), "wall_us_per_batch": round(1e6 * wall / (24 * 6), 2),
           "GBps": round(24 * 6 * (64 << 20) / wall / 1e9, 1)}
    for kn in ("k_plan", "k_plan_pre"):
        if kn not in prof or not PROF:
            continue
        n, ms = prof[kn][0] - base.get(kn, (0, 0))[0], prof[kn][1] - base.get(kn, (0, 0.0))[1]
        if n:
            row[kn + "_us"] = round(1e3 * ms / n, 2)
            row[kn + "_launches"] = n
    print(json.dumps(row), flush=True)
    rows.append(row)
    d.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/plan_probe.json", "w"), indent=1)
