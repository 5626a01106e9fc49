"""Mutation fuzz of the table-copy path against the oracle (emulator or GPU): python tools/copy_fuzz.py [iterations] [seed]
Every iteration takes generated COPY rows, mutates a few bytes (specials, invalid UTF-8, deletions, insertions) in a few rows and
compares error (code, kind, description, row) and the arena of the rows before it with oracle/ — the rows -> arena kernel must never
accept what the reference rejects, and a batch it gives up on must come out of the row -> frame rewrite exactly as before."""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import synth
from etl_amd.decoder import Decoder
from oracle import oracle
from tests import test_oracle_copy as K

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
TABLES = [(synth.COPY_COLS, lambda n, s: synth.copy_rows(n, s)),
          ([(c, K.TEXT, True, 0) for c in "abc"], lambda n, s: text_rows(n, s)),
          ([("c%d" % i, K.TEXT, True, 0) for i in range(20)], lambda n, s: text_rows(n, s, 20))]   # more than 16 columns: the WIDE instantiation
TOKENS = ["a", "b", "\\\\", "\\t", "\\n", "\\N", "é", "中", " ", "\\\t", "\\\n", "N", "xyz" * 5]   # every token is a whole character or a whole escape


def text_rows(n, s, nf=3):
    r = random.Random(s)
    return [("\t".join(r.choice(["\\N", "", "".join(r.choice(TOKENS) for _ in range(r.randrange(1, 14)))]) if r.random() < 0.25
                       else "".join(r.choice(TOKENS) for _ in range(r.randrange(0, 14))) for _ in range(nf)) + "\n").encode() for _ in range(n)]


SPECIALS = [b"\t", b"\n", b"\\", b"\\N", b"\\\\", b"\xff", b"\xc3", b"\xa9", b"", b"N", b"\\\t", b"\\\n", b"\xe4\xb8", b"0", b"x"]
bad = 0
direct = frames = 0
for it in range(iters):
    cols, gen = TABLES[it % len(TABLES)]
    rows = list(gen(rng.randrange(1, 200), rng.randrange(1 << 20)))
    benign = it % len(TABLES) != 0 and it % 2 == 1   # every other batch of the all-text table: insertions that keep the rows valid (pairs that
                                            # start with a backslash, plain and multi-byte characters), many of them — batches the one-kernel path keeps
    if benign:
        for _ in range(rng.randrange(1, 60)):
            r = rng.randrange(len(rows))
            f = rows[r][:-1].split(b"\t")   # raw tabs are separators here (escaped ones are "\\" + TAB: the split may cut behind the backslash,
            k = rng.randrange(len(f))        # then the insertion lands between them and unescapes the tab into a separator: still a row, maybe a bad one)
            ins = rng.choice([b"\\\\", b"\\t", b"\\n", b"\\N", b"\\x", b"z", b"\xc3\xa9", b"\\\xc3\xa9", b"NN", b"\\b"])
            at = rng.randrange(len(f[k]) + 1)
            f[k] = f[k][:at] + ins + f[k][at:]
            rows[r] = b"\t".join(f) + b"\n"
    for _ in range(0 if benign else rng.randrange(0, 4)):
        r = rng.randrange(len(rows))
        row = bytearray(rows[r])
        k = rng.random()
        pos = rng.randrange(len(row) + 1)
        if k < 0.5:
            row[pos:pos] = rng.choice(SPECIALS)
        elif k < 0.8 and row:
            pos = min(pos, len(row) - 1)
            row[pos:pos + 1] = rng.choice(SPECIALS)
        elif row:
            del row[min(pos, len(row) - 1)]
        rows[r] = bytes(row)
    o, d = oracle.Oracle(), Decoder(0)
    for t in (o, d):
        t.schema_put(42, 0, cols)
    so = o.table_ready(42, 0, [1] * len(cols), [1 if c[3] else 0 for c in cols])
    sd = d.table_ready(42, 0, [1] * len(cols), [1 if c[3] else 0 for c in cols])
    if os.environ.get("COPY_FUZZ_TRACE"):
        open(os.environ["COPY_FUZZ_TRACE"], "wb").write(repr((it, cols, rows)).encode())
    buf = np.frombuffer(b"".join(rows), dtype=np.uint8)
    offs = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint32)
    rb, gb = o.copy_decode(so, buf, offs), d.copy_decode(sd, buf, offs)
    e = gb.error
    got = (e.code, e.kind, e.description, e.frame_index) if e else (0, 0, "", -1)
    want = (rb.err_code, rb.err_kind, rb.err_desc, rb.err_frame)
    diff = rb.host_batch().diff(gb.host())
    dc = d.debug_copy()
    direct += dc["direct"]; frames += dc["frames"]
    if got != want or diff or (want[0] != 0 and dc["direct"]):
        bad += 1
        print("MISMATCH it", it, "want", want, "got", got, "diff", diff[:3], dc)
        open("/tmp/copy_fuzz_case_%d.bin" % it, "wb").write(repr((cols, rows)).encode())
        if bad > 5:
            break
    d.close()
print("iterations", it + 1, "mismatches", bad, "direct", direct, "frames", frames)
