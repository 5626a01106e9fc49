"""Re-creates inputs of tools/simt_fuzz.py's boundary-scan worker (seed, iterations ...) and runs etlg_scan_boundaries on them repeatedly:
python tools/scan_repro.py seed it [it ...]   — prints want / got frame counts per repetition and the first differing offset."""
import os
import random
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd.decoder import Decoder
from tests.test_gpu_scan import ref_scan

seed = int(sys.argv[1])
wanted = sorted(int(a) for a in sys.argv[2:])
rng = random.Random(seed)


def payload(n):
    k = rng.random()
    if k < 0.3:
        return bytes(rng.getrandbits(8) for _ in range(n))
    if k < 0.6:
        return bytes(rng.choice(b"d\x00\x00\x01\x10w") for _ in range(n))
    return (b"d" + struct.pack(">I", rng.choice([4, 5, 17, 60, 200, 4000])) + b"w") * (n // 6 + 1)


def gen():
    parts = []
    total = rng.choice([0, 1, 4, 5, 300, 5000, 9000, 40000, 150000])
    size = 0
    while size < total:
        n = rng.choice([0, 1, 20, 108, 108, 108, 500, 3000, 9000, 20000, 70000])
        body = payload(n)[:n]
        fr = b"d" + struct.pack(">I", len(body) + 4) + body
        k = rng.random()
        if k < 0.03:
            fr = fr[:rng.randrange(1, len(fr) + 1)]
        elif k < 0.05:
            fr = bytes([rng.getrandbits(8)]) + fr[1:]
        elif k < 0.07:
            fr = fr[:1] + struct.pack(">I", rng.choice([0, 3, 2**31, 2**32 - 1, len(body) + 5])) + fr[5:]
        parts.append(fr)
        size += len(fr)
    buf = np.frombuffer(b"".join(parts), dtype=np.uint8)
    if rng.random() < 0.3 and len(buf) > 3:
        buf = buf[:rng.randrange(len(buf))]
    return buf


bufs = {}
it = 0
while wanted and it <= wanted[-1]:
    b = gen()
    if it in wanted:
        bufs[it] = b.copy()
    it += 1
os.makedirs("gpurun_out", exist_ok=True)
for it, buf in bufs.items():
    np.save(f"gpurun_out/scan_repro_{seed}_{it}.npy", buf)
    want = ref_scan(buf)
    print("it", it, "len", len(buf), "want frames", len(want) - 1, "offsets", want[:8], flush=True)
    for mode in ("fresh", "same"):
        dec = Decoder(0) if mode == "same" else None
        res = []
        for rep in range(10):
            d = dec or Decoder(0)
            got = d.scan_boundaries(buf)
            ok = len(got) == len(want) and np.array_equal(got, want)
            res.append("ok" if ok else f"BAD({len(got) - 1}: {list(got[:8])})")
            if not dec:
                d.close()
        if dec:
            print("   reruns/seq", dec.debug_scan() if hasattr(dec, "debug_scan") else "", flush=True)
            dec.close()
        print("  ", mode, res, flush=True)
