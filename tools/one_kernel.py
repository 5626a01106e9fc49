"""One workload, a few launches of its decode kernel (for counter passes): python tools/one_kernel.py cfg3|cfg5|wide70|copy|copy_clean|nosidecar|finish [launches]
(the batch shapes are bench.py's: 64 MiB; wide70 44 000 rows; copy 400 000 rows)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
d = Decoder(0)
if wl in ("copy", "copy_clean"):
    base = synth.copy_rows(20000, 1, clean=True) if wl == "copy_clean" else synth.copy_rows(20000, 1)
    rows = base * 20
    d.schema_put(42, 0, synth.COPY_COLS)
    slot = d.table_ready(42, 0, [1] * 10, [1] + [0] * 9)
    buf = np.frombuffer(b"".join(rows), dtype=np.uint8)
    offs = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint32)
    tb = torch.from_numpy(buf.copy()).cuda(); to = torch.from_numpy(offs.view(np.int32).copy()).cuda()
    torch.cuda.synchronize()
    for it in range(n):
        b = d.copy_decode_device(slot, tb.data_ptr(), tb.numel(), to.data_ptr(), len(rows)); assert b.rc == 0, b.error; b.close()
    print(d.debug_paths())
    d.close()
    sys.exit(0)
if wl in ("wide70", "finish"):
    buf, offs = synth.type_matrix_stream(44000, mix=True)
    synth.type_matrix_register(d)
    tb = torch.from_numpy(buf.copy()).cuda(); to = torch.from_numpy(offs.astype(np.uint32).view(np.int32).copy()).cuda()
    for it in range(n):
        b = d.decode_device(tb.data_ptr(), tb.numel(), to.data_ptr(), len(offs) - 1, abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL | (abi.F_FINISH_CELLS if wl == "finish" else 0))
        assert b.rc == 0, b.error
        b.close()
    print(d.debug_paths(), d.debug_rows())
    d.close()
    sys.exit(0)
w = getattr(synth, "cfg2" if wl == "nosidecar" else wl)()
ready = not w.cfg.emit_relations
w.register(d, ready=ready)
for it in range(n):
    buf, offs = w.fill(64 << 20)
    tb = torch.from_numpy(buf.copy()).cuda(); to = torch.from_numpy(offs.astype(np.uint32).view(np.int32).copy()).cuda()
    if wl == "nosidecar":
        b = d.decode_device(tb.data_ptr(), tb.numel(), None, 0, abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL)
    else:
        b = d.decode_device(tb.data_ptr(), tb.numel(), to.data_ptr(), len(offs) - 1, abi.F_OUTPUT_ON_DEVICE | (abi.F_NO_CONTROL if ready else 0))
    assert b.rc == 0, b.error
    b.close()
print(d.debug_paths(), d.debug_rows())
d.close()
