"""Fuzz of the ASYNC table-copy form (etlg_copy_decode with ETLG_F_ASYNC) against the oracle, emulator or GPU:
    python tools/copy_async_fuzz.py [seconds=20] [seed=1]
One context per table for the whole run (buffers, side-input sets, result ring and descriptor buffers go through their rotations);
every round enqueues one to seven batches of generated COPY rows — a few bytes mutated in some (specials, invalid UTF-8, deletions:
rows the reference rejects, rows the one-kernel path hands to the frame rewrite) — without a sync in between, now and then with a WAL
transaction decoded in the middle (which must finish the copy batches first and leave its own state alone), then syncs them in issue
order: error (code, kind, description, row), every arena byte before it and the payload metadata as oracle/ has them."""
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
from oracle import oracle
from tests import pgwire as W
from tests import scenarios as SC
from tests import test_oracle_copy as K

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
TOKENS = ["a", "b", "\\\\", "\\t", "\\n", "\\N", "é", "中", " ", "\\\t", "\\\n", "N", "xyz" * 5]
SPECIALS = [b"\t", b"\n", b"\\", b"\\N", b"\\\\", b"\xff", b"\xc3", b"\xa9", b"", b"N", b"\\\t", b"\\\n", b"\xe4\xb8", b"0", b"x"]


def text_rows(n, s, nf=3):
    r = random.Random(s)
    return [("\t".join(r.choice(["\\N", "", "".join(r.choice(TOKENS) for _ in range(r.randrange(1, 14)))]) if r.random() < 0.25
                       else "".join(r.choice(TOKENS) for _ in range(r.randrange(0, 14))) for _ in range(nf)) + "\n").encode() for _ in range(n)]


TABLES = [(synth.COPY_COLS, lambda n, s: synth.copy_rows(n, s)),
          ([(c, K.TEXT, True, 0) for c in "abc"], lambda n, s: text_rows(n, s)),
          ([("c%d" % i, K.TEXT, True, 0) for i in range(20)], lambda n, s: text_rows(n, s, 20))]
ASYNC = abi.F_ASYNC | abi.F_OUTPUT_ON_DEVICE
MIXED_VALS = [r.decode() for r in [b"1", b"2", b"t", b"1", b"x", b"y", b"2024-01-01 00:00:00+00", b"123e4567-e89b-12d3-a456-426614174000", b"1.5", b"\\x00"]]

ctxs = []
for cols, gen in TABLES:
    o, d = oracle.Oracle(), Decoder(0)
    for t in (o, d):
        t.schema_put(42, 0, cols)
    so = o.table_ready(42, 0, [1] * len(cols), [1 if c[3] else 0 for c in cols])
    sd = d.table_ready(42, 0, [1] * len(cols), [1 if c[3] else 0 for c in cols])
    assert so == sd
    for t in (o, d):
        t.table_state(42, abi.TS_READY)
    ctxs.append((cols, gen, o, d, so))

t0 = time.time()
rounds = batches = bad = errs = wal = 0
lsn = 0x2000000
while time.time() - t0 < seconds and bad < 5:
    ti = rng.randrange(len(TABLES))
    cols, gen, o, d, slot = ctxs[ti]
    todo = []
    for _ in range(rng.randrange(1, 8)):
        rows = list(gen(rng.choice([1, 3, 60, 64, 65, 200, 700, 2500]), rng.randrange(1 << 20)))
        for _ in range(rng.choice([0, 0, 0, 1, 2])):
            r = rng.randrange(len(rows))
            row = bytearray(rows[r])
            pos = rng.randrange(len(row) + 1)
            k = rng.random()
            if k < 0.5:
                row[pos:pos] = rng.choice(SPECIALS)
            elif k < 0.8 and row:
                pos = min(pos, len(row) - 1)
                row[pos:pos + 1] = rng.choice(SPECIALS)
            elif row:
                del row[min(pos, len(row) - 1)]
            rows[r] = bytes(row)
        todo.append((np.frombuffer(b"".join(rows), dtype=np.uint8), np.cumsum([0] + [len(r) for r in rows]).astype(np.uint32)))
    cut = rng.randrange(len(todo) + 1) if (ti == 0 and rng.random() < 0.3) else -1
    inflight = []
    for k, (buf, offs) in enumerate(todo):
        if k == cut:   # a WAL transaction on the same context while copy batches are in flight
            s = SC.txn([W.insert(42, MIXED_VALS)], final=lsn + 0x800, lsn0=lsn)
            lsn += 0x1000
            wb = np.frombuffer(s.bytes(), dtype=np.uint8)
            r2, g2 = o.decode(wb, s.offsets), d.decode(wb, s.offsets)
            wd = r2.host_batch().diff(g2.host()) if (r2.err_code == 0 and g2.rc == 0) else []
            if r2.err_code != 0 or g2.rc != 0 or wd:
                bad += 1
                print("MISMATCH (WAL batch between copy batches) round", rounds, "k", k, "of", len(todo), "oracle", r2.err_code, r2.err_desc, "device", g2.rc, g2.error.description if g2.error else "", wd[:4], flush=True)
            wal += 1
        # (one call in six is synchronous: it finishes the batches in flight first and comes back finished itself)
        inflight.append(d.copy_decode(slot, buf, offs, flags=ASYNC if rng.random() < 0.84 else 0))
    for k, ((buf, offs), g) in enumerate(zip(todo, inflight)):
        rb = o.copy_decode(slot, buf, offs)
        g.sync()
        e = g.error
        got = (e.code, e.kind, e.description, e.frame_index) if e else (0, 0, "", -1)
        want = (rb.err_code, rb.err_kind, rb.err_desc, rb.err_frame)
        diff = [] if got != want else rb.host_batch().diff(g.host())
        done = rb.err_frame if rb.err_code else len(offs) - 1
        pay_ok = g.view().payload_bytes[0] == int(offs[done]) - int(offs[0])
        errs += want[0] != 0
        batches += 1
        if got != want or diff or not pay_ok:
            bad += 1
            print("MISMATCH round", rounds, "batch", k, "of", len(todo), "table", ti, "want", want, "got", got, "diff", diff[:3], "payload ok", pay_ok, flush=True)
        g.close()
    rounds += 1
dc = [c[3].debug_copy() for c in ctxs]
for c in ctxs:
    c[3].close()
print("rounds", rounds, "batches", batches, "with errors", errs, "WAL batches in between", wal, "mismatches", bad,
      "direct", sum(x["direct"] for x in dc), "frames", sum(x["frames"] for x in dc), flush=True)
sys.exit(1 if bad else 0)
