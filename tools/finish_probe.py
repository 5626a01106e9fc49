"""The finish pass on the type-matrix table: wall time of etlg_batch_finish_cells per batch (python tools/finish_probe.py [rows]); under
rocprofv3 --kernel-trace --stats the kernels' own times."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
nrows = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
buf, offs = synth.type_matrix_stream(nrows, mix=True)
tb = torch.from_numpy(buf.copy()).cuda(); to = torch.from_numpy(offs.astype(np.uint32).view(np.int32).copy()).cuda()
d = Decoder(0)
synth.type_matrix_register(d)
FL = abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL
for it in range(6):
    b = d.decode_device(tb.data_ptr(), tb.numel(), to.data_ptr(), len(offs) - 1, FL)
    assert b.rc == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = b.finish_cells()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"batch {len(buf)} B, {len(offs) - 1} frames: finish {1e3 * (t1 - t0):.3f} ms; typed {st.arrays_typed}, floats {st.floats_settled}, left {st.left_deferred}, heap +{st.heap_bytes_added}", flush=True)
    b.close()
d.close()
