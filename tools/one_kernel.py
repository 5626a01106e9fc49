"""One workload, a few launches of its decode kernel (for counter passes): python tools/one_kernel.py cfg3|cfg5|wide70 [launches]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
w = getattr(synth, wl)()
d = Decoder(0)
ready = not w.cfg.emit_relations
w.register(d, ready=ready)
for it in range(n):
    buf, offs = w.fill(64 << 20)
    tb = torch.from_numpy(buf.copy()).cuda(); to = torch.from_numpy(offs.astype(np.uint32).view(np.int32).copy()).cuda()
    b = d.decode_device(tb.data_ptr(), tb.numel(), to.data_ptr(), len(offs) - 1, abi.F_OUTPUT_ON_DEVICE | (abi.F_NO_CONTROL if ready else 0))
    assert b.rc == 0, b.error
    b.close()
print(d.debug_paths(), d.debug_rows())
d.close()
