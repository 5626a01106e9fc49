# Table-copy splitter A/B on two kinds of rows: python tools/bench_copy2.py   (ETLG_COPY_KERNEL=0 -> lane-per-row splitter)
import os, sys, time, random
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
def clean_rows(n, seed):
    rng = random.Random(seed)
    alphabet = "abcdefghijklmnopqrstuvwxyz ABCDEFGHIJ0123456789,;.-"
    rows = []
    for i in range(n):
        txt = "".join(rng.choice(alphabet) for _ in range(rng.randint(0, 80)))
        f = [str(i), str(rng.randint(-2**31, 2**31 - 1)), rng.choice("tf"), rng.choice(["0", "-12.5", "123456789.000100", "NaN", "1e5", "0.000012"]),
             txt, "\\N" if rng.random() < 0.3 else txt[:10],
             "2024-0%d-1%d 0%d:30:15.%06d+0%d" % (rng.randint(1, 9), rng.randint(0, 9), rng.randint(0, 9), rng.randint(0, 999999), rng.randint(0, 9)),
             "%08x-1111-2222-3333-%012x" % (rng.getrandbits(32), rng.getrandbits(48)), rng.choice(["1.5", "-0.25", "1e300", "3.141592653589793"]),
             "\\\\x" + "".join("%02x" % rng.getrandbits(8) for _ in range(rng.randint(0, 20)))]
        rows.append(("\t".join(f) + "\n").encode())
    return rows
for name, base in (("escape-heavy (bench.py's rows)", synth.copy_rows(20000, 1)), ("clean text", clean_rows(20000, 1))):
    rows = base * 20
    d = Decoder(0)
    d.schema_put(42, 0, synth.COPY_COLS)
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
    print({"rows": name, "kernel": os.environ.get("ETLG_COPY_KERNEL", "1"), "bytes": len(buf), "GB/s": round(K * len(buf) / (t1 - t0) / 1e9, 2), "kernel_us": prof})
    d.close()
