"""k_rows against k_cells on the variable-length workloads in one process: cfg3 (one 64 MiB batch, NO_CONTROL), cfg5 (consecutive 64 MiB
batches, default flags), wide70 — kernel time from the library's HIP events (etlg_ctx_profile), the kernel chosen through ETLG_ROWS /
ETLG_FUSED_KERNEL. usage: python tools/rows_ab.py [label]   (measurement tool, not product)"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import abi, synth
from etl_amd.decoder import Decoder

label = sys.argv[1] if len(sys.argv) > 1 else "default"
row = {"label": label, "env": {k: os.environ[k] for k in ("ETLG_ROWS", "ETLG_FUSED_KERNEL", "ETLG_FUSED_DBG", "ETLG_ROWS_NW", "ETLG_ROWS_CF") if k in os.environ}}


def dev(buf, offs):
    return torch.from_numpy(buf.copy()).cuda(), torch.from_numpy(offs.astype(np.uint32).view(np.int32).copy()).cuda()


def per_launch(prof, base):
    out = {}
    for k, (n, ms) in prof.items():
        n0, ms0 = base.get(k, (0, 0.0))
        if n > n0:
            out[k] = round(1e3 * (ms - ms0) / (n - n0), 1)
    return out


w = synth.cfg3()
buf, offs = w.fill(64 << 20)
tb, to = dev(buf, offs)
d = Decoder(0)
synth.cfg3().register(d, ready=True)
d.profile(True)
base = {}
for it in range(4 + 16):
    if it == 4:
        base = d.profile_read()
    b = d.decode_device(tb.data_ptr(), tb.numel(), to.data_ptr(), len(offs) - 1, abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL)
    assert b.rc == 0, b.error
    b.close()
row["cfg3"] = per_launch(d.profile_read(), base)
row["cfg3_paths"] = {**d.debug_paths(), **d.debug_rows()}
if os.environ.get("ETLG_FUSED_DBG"):
    import ctypes as C
    out = (C.c_ulonglong * 12)()
    d.L.etlg_ctx_debug_times(d.h, out)
    row["cfg3_phase_cycles_sum_over_sampled_tiles"] = [int(x) for x in out]
d.close()

w = synth.cfg5()
d = Decoder(0)
w.register(d, ready=False)
d.profile(True)
batches = [dev(*w.fill(64 << 20)) for _ in range(8)]
base = {}
for it, (tb, to) in enumerate(batches):
    if it == 2:
        base = d.profile_read()
    b = d.decode_device(tb.data_ptr(), tb.numel(), to.data_ptr(), to.numel() - 1, abi.F_OUTPUT_ON_DEVICE)
    assert b.rc == 0, b.error
    b.close()
row["cfg5"] = per_launch(d.profile_read(), base)
row["cfg5_paths"] = {**d.debug_paths(), **d.debug_rows()}
d.close()
del batches
print(json.dumps(row), flush=True)
