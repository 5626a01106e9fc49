"""k_cells on cfg3, A/B of runtime knobs in ONE process (boxes differ by up to 10 %): kernel time from the library's HIP events.
usage: python tools/cells_ab.py name:ENV=V,ENV2=V ...   e.g.  rot: norot:ETLG_FUSED_DBG=65536     (each variant measured twice, interleaved)"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
variants = [a for a in sys.argv[1:] if not a.startswith("wl=")] or ["default:"]
WL = ([a[3:] for a in sys.argv[1:] if a.startswith("wl=")] or ["cfg3"])[0]
MK = getattr(synth, WL)
w = MK()
buf, offs = w.fill(64 << 20)
tb = torch.from_numpy(buf.copy()).cuda()
to = torch.from_numpy(offs.astype(np.uint32).view(np.int32).copy()).cuda()
nf = len(offs) - 1
KNOBS = set()
for v in variants:
    for kv in filter(None, v.partition(":")[2].split(",")):
        KNOBS.add(kv.split("=")[0])
for rep in range(2):
    for v in variants:
        name, _, envs = v.partition(":")
        for k in KNOBS:
            os.environ.pop(k, None)
        for kv in filter(None, envs.split(",")):
            a, b = kv.split("=")
            os.environ[a] = b
        d = Decoder(0)
        MK().register(d, ready=True)
        d.profile(True)
        base = {}
        for it in range(4 + 16):
            if it == 4:
                base = d.profile_read()
            b = d.decode_device(tb.data_ptr(), tb.numel(), to.data_ptr(), nf, abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL)
            assert b.rc == 0, b.error
            b.close()
        prof = d.profile_read()
        row = {"workload": WL, "variant": name, "env": envs, "rep": rep}
        for k, (n, ms) in prof.items():
            n0, ms0 = base.get(k, (0, 0.0))
            if n > n0:
                row[k + "_us"] = round(1e3 * (ms - ms0) / (n - n0), 1)
        row["paths"] = d.debug_paths()
        print(json.dumps(row), flush=True)
        d.close()
