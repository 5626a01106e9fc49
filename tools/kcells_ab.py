"""k_cells on its three workloads in one process: cfg3 (one 64 MiB batch, NO_CONTROL), cfg5 (consecutive 64 MiB batches of one stream,
default flags: the control path), table-copy rows (ordinary / escape-heavy) — kernel time from the library's HIP events. The library
is whatever ETLG_LIB_PATH names, so variants are compared by running this once per library in ONE gpurun call (boxes differ by ~10 %).
usage: python tools/kcells_ab.py [label]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import abi, synth
from etl_amd.decoder import Decoder

label = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("ETLG_LIB_PATH", "default")
row = {"lib": label}


def dev(buf, offs):
    return torch.from_numpy(buf.copy()).cuda(), torch.from_numpy(offs.astype(np.uint32).view(np.int32).copy()).cuda()


def per_launch(prof, base):
    out = {}
    for k, (n, ms) in prof.items():
        n0, ms0 = base.get(k, (0, 0.0))
        if n > n0:
            out[k] = round(1e3 * (ms - ms0) / (n - n0), 1)
    return out


# cfg3
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
    assert b.rc == 0 or os.environ.get("PROBE_NO_ASSERT"), b.error
    b.close()
row["cfg3"] = per_launch(d.profile_read(), base)
d.close()

# cfg5: consecutive batches, default flags
w = synth.cfg5()
d = Decoder(0)
w.register(d, ready=False)
d.profile(True)
batches = [dev(*w.fill(64 << 20)) + (None,) for _ in range(10)]
base = {}
for it, (tb, to, _) in enumerate(batches):
    if it == 2:
        base = d.profile_read()
    b = d.decode_device(tb.data_ptr(), tb.numel(), to.data_ptr(), to.numel() - 1, abi.F_OUTPUT_ON_DEVICE)
    assert b.rc == 0 or os.environ.get("PROBE_NO_ASSERT"), b.error
    b.close()
row["cfg5"] = per_launch(d.profile_read(), base)
row["cfg5_paths"] = d.debug_paths()
d.close()
del batches

# table copy
for clean in (True, False):
    rows = synth.copy_rows(20000, 1, clean=clean) * 20
    buf = np.frombuffer(b"".join(rows), dtype=np.uint8)
    offs = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint32)
    tb, to = dev(buf, offs)
    d = Decoder(0)
    d.schema_put(42, 0, synth.COPY_COLS)
    slot = d.table_ready(42, 0, [1] * 10, [1] + [0] * 9)
    d.profile(True)
    base = {}
    for it in range(2 + 8):
        if it == 2:
            base = d.profile_read()
        d.copy_decode_device(slot, tb.data_ptr(), tb.numel(), to.data_ptr(), len(rows)).close()
    row["copy_clean" if clean else "copy_escapes"] = per_launch(d.profile_read(), base)
    d.close()
print(json.dumps(row), flush=True)
