"""Kernel times of ONE build of the library (ETLG_LIB_PATH: a build/variants/*.so of tools/build_variant.py, or the shipped one) on the
bench's batch shapes, from the library's HIP events (etlg_ctx_profile): cfg2 (k_plan_pre + k_plan), cfg3 / wide70 (k_rows), table copy
escape-heavy / ordinary (k_copy_cells). usage: python tools/variant_ab.py label [cfg2,cfg3,wide70,copy]   (measurement tool, not product)"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import abi, synth
from etl_amd.decoder import Decoder

label = sys.argv[1] if len(sys.argv) > 1 else "shipped"
legs = (sys.argv[2] if len(sys.argv) > 2 else "cfg2,cfg3,wide70,copy").split(",")
row = {"label": label}


def per_launch(prof, base):
    return {k: round(1e3 * (ms - base.get(k, (0, 0.0))[1]) / (n - base.get(k, (0, 0.0))[0]), 1) for k, (n, ms) in prof.items() if n > base.get(k, (0, 0.0))[0]}


def timed(d, call, warm=4, reps=16):
    d.profile(True)
    base = {}
    for it in range(warm + reps):
        if it == warm:
            base = d.profile_read()
        b = call()
        assert b.rc == 0, b.error
        b.close()
    return per_launch(d.profile_read(), base)


def wal(w, flags, nbytes=64 << 20):
    buf, offs = w.fill(nbytes)
    tb = torch.from_numpy(buf.copy()).cuda(); to = torch.from_numpy(offs.astype(np.uint32).view(np.int32).copy()).cuda()
    d = Decoder(0)
    w.register(d, ready=True)
    out = timed(d, lambda: d.decode_device(tb.data_ptr(), tb.numel(), to.data_ptr(), len(offs) - 1, flags))
    d.close()
    return out


FL = abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL
if "cfg2" in legs:
    row["cfg2"] = wal(synth.cfg2(), FL)
if "cfg3" in legs:
    row["cfg3"] = wal(synth.cfg3(), FL)
if "wide70" in legs:
    buf, offs = synth.type_matrix_stream(44000, mix=True)
    d = Decoder(0)
    synth.type_matrix_register(d)
    tb = torch.from_numpy(buf.copy()).cuda(); to = torch.from_numpy(offs.astype(np.uint32).view(np.int32).copy()).cuda()
    row["wide70"] = timed(d, lambda: d.decode_device(tb.data_ptr(), tb.numel(), to.data_ptr(), len(offs) - 1, FL))
    d.close()
if "copy" in legs:
    for name, clean in (("copy", False), ("copy_clean", True)):
        rows = (synth.copy_rows(20000, 1, clean=True) if clean else synth.copy_rows(20000, 1)) * 20
        d = Decoder(0)
        d.schema_put(42, 0, synth.COPY_COLS)
        slot = d.table_ready(42, 0, [1] * 10, [1] + [0] * 9)
        buf = np.frombuffer(b"".join(rows), dtype=np.uint8)
        offs = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint32)
        tb = torch.from_numpy(buf.copy()).cuda(); to = torch.from_numpy(offs.view(np.int32).copy()).cuda()
        row[name] = timed(d, lambda: d.copy_decode_device(slot, tb.data_ptr(), tb.numel(), to.data_ptr(), len(rows)))
        d.close()
print(json.dumps(row), flush=True)
