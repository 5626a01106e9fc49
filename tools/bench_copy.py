# Table-copy throughput (not the BASELINE metric): python tools/bench_copy.py [rows]
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from etl_amd import abi
from etl_amd.decoder import Decoder
from tests.test_gpu_copy import _gen_rows, GEN_COLS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
rows = _gen_rows(20000, 1) * (n // 20000)
d = Decoder(0)
d.schema_put(42, 0, GEN_COLS)
slot = d.table_ready(42, 0, [1] * 10, [1] + [0] * 9)
buf = np.frombuffer(b"".join(rows), dtype=np.uint8)
offs = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint32)
tb = torch.from_numpy(buf.copy()).cuda(); to = torch.from_numpy(offs.view(np.int32).copy()).cuda()
torch.cuda.synchronize()
for _ in range(3):
    d.copy_decode_device(slot, tb.data_ptr(), tb.numel(), to.data_ptr(), len(rows)).close()
d.profile(True)
t0 = time.perf_counter(); K = 10
for _ in range(K):
    b = d.copy_decode_device(slot, tb.data_ptr(), tb.numel(), to.data_ptr(), len(rows)); assert b.rc == 0; b.close()
torch.cuda.synchronize(); t1 = time.perf_counter()
prof = {k: round(1000 * ms / max(c, 1), 1) for k, (c, ms) in d.profile_read().items() if c}
print({"rows": len(rows), "bytes": len(buf), "GB/s": round(K * len(buf) / (t1 - t0) / 1e9, 2), "rows/s": round(K * len(rows) / (t1 - t0)), "kernel_us": prof})
