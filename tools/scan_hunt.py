"""Boundary-scan byte-soup fuzz in ONE process (the worker of tools/simt_fuzz.py), with context around every disagreement:
python tools/scan_hunt.py seed seconds   — prints the previous input's size, the scan counters before / after, and whether the same
input passes when it is scanned again at once (same context) and on a fresh context."""
import os
import random
import struct
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd.decoder import Decoder
from tests.test_gpu_scan import ref_scan

seed, budget = int(sys.argv[1]), float(sys.argv[2])
rng = random.Random(seed)
dec = Decoder(0)


def payload(n):
    k = rng.random()
    if k < 0.3:
        return bytes(rng.getrandbits(8) for _ in range(n))
    if k < 0.6:
        return bytes(rng.choice(b"d\x00\x00\x01\x10w") for _ in range(n))
    return (b"d" + struct.pack(">I", rng.choice([4, 5, 17, 60, 200, 4000])) + b"w") * (n // 6 + 1)


t0 = time.time()
it = bad = 0
prev_len = -1
while time.time() - t0 < budget:
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
    c0 = dec.debug_scan()
    got = dec.scan_boundaries(buf)
    c1 = dec.debug_scan()
    want = ref_scan(buf)
    if len(got) != len(want) or not np.array_equal(got, want):
        bad += 1
        again = dec.scan_boundaries(buf)
        d2 = Decoder(0)
        fresh = d2.scan_boundaries(buf)
        d2.close()
        nd = 0
        while nd < min(len(got), len(want)) and got[nd] == want[nd]:
            nd += 1
        print("MISMATCH it", it, "len", len(buf), "tiles", (len(buf) + 8191) // 8192, "prev_len", prev_len, "want", len(want) - 1, "got", len(got) - 1,
              "first diff at frame", nd, "want", list(want[nd:nd + 3]), "got", list(got[nd:nd + 3]), "counters", c0, "->", c1,
              "again ok" if np.array_equal(again, want) else "again BAD", "fresh ok" if np.array_equal(fresh, want) else "fresh BAD", flush=True)
        if bad <= 3:
            np.save(f"gpurun_out/scan_hunt_{seed}_{it}.npy", buf)
    prev_len = len(buf)
    it += 1
print("DONE iterations", it, "mismatches", bad)
