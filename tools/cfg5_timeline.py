"""cfg5 ASYNC chain, host-side timeline: when every etlg_decode call returns and when every batch is synced (ms since the first call).
python tools/cfg5_timeline.py [npool]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
npool = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
w = synth.cfg5()
pool = [w.fill(64 << 20) for _ in range(npool)]
items = bench.to_device(pool, dev)
dec = Decoder(0)
FL = abi.F_OUTPUT_ON_DEVICE | abi.F_ASYNC
for rep in range(3):
    for t in w.tables:
        dec.table_forget(t["rel_id"])
    dec.reset_stream_state()
    w.register(dec, ready=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bs, issued, synced = [], [], []
    for tb, to, nb, nf in items:
        bs.append(dec.decode_device(tb.data_ptr(), nb, to.data_ptr(), nf, FL))
        issued.append((time.perf_counter() - t0) * 1e3)
    for b in bs:
        assert b.sync() == 0
        synced.append((time.perf_counter() - t0) * 1e3)
    for b in bs:
        b.close()
    torch.cuda.synchronize()
    print("rep", rep, "total %.2f ms" % ((time.perf_counter() - t0) * 1e3), "paths", dec.debug_paths())
    print("  issued", " ".join("%.2f" % x for x in issued))
    print("  synced", " ".join("%.2f" % x for x in synced))
dec.close()
