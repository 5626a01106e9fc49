"""One workload decoded a few times, for `rocprofv3 --pc-sampling-beta-enabled ...` (tools/run_pcsample.sh): cfg3 (k_cells) by default,
cfg5 / copy with an argument."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
w = getattr(synth, wl)()
d = Decoder(0)
w.register(d, ready=(wl != "cfg5"))
buf, offs = w.fill(64 << 20)
tb = torch.from_numpy(buf.copy()).cuda()
to = torch.from_numpy(offs.astype(np.uint32).view(np.int32).copy()).cuda()
fl = abi.F_OUTPUT_ON_DEVICE | (abi.F_NO_CONTROL if wl != "cfg5" else 0)
for it in range(n):
    b = d.decode_device(tb.data_ptr(), tb.numel(), to.data_ptr(), len(offs) - 1, fl)
    assert b.rc == 0, b.error
    b.close()
print("done", d.debug_paths())
d.close()
