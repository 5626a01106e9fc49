"""Long ASYNC chains on ONE context against the oracle (emulator or GPU): python tools/async_long_fuzz.py [seconds=120] [seed=1]
What tools/async_fuzz.py does not reach: its rounds use a fresh context for at most 40 batches, so the result ring (32 blocks) barely
laps and a batch that is decoded again never meets a re-used block. Here a round is 80-200 cfg2 batches cut at arbitrary frames on one
context, a random window of them in flight, and in a few of them an UPDATE the fixed-width plan does not cover (the batch is decoded
again by the generic kernel when it is synced, and so are the batches that were in flight behind it); every batch against the oracle,
every field — and the chain has to heal: second attempts stay within a window's worth per spliced batch.
(Found-by-design targets: the two defects of round 5's second session)"""
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import synth
from etl_amd.decoder import Decoder
from oracle import oracle
from tests import pgwire as W
from tests.test_gpu_async import FLAGS, DevBufs, _cut

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mode = sys.argv[3] if len(sys.argv) > 3 else "cfg2"    # "nosidecar" / "mixed": (some) batches without the offsets sidecar; "cfg5": the DDL workload with default flags — the pipelined control path (pre-pass ring, two scratch sets) over many laps
if mode in ("cfg5", "ddl", "ddl_fixed"):
    from tests.test_gpu_async import FLAGS_DEFAULT
    t_end = time.time() + seconds
    rounds = batches = bad = 0
    while time.time() < t_end or rounds == 0:
        rng = random.Random(seed + rounds)
        if mode == "ddl":   # DDL message -> Relation every few transactions: EVERY batch carries control frames, the chain stays on the pipelined control path
            w = synth.Workload([synth.table_fixed(), synth.table_w8(), synth.table_mixed()], 0xE7D0000 + seed * 1000 + rounds, rows_per_txn=rng.choice([3, 5, 12]),
                               mix=(60, 30, 10), upd_key=10, upd_toast=5, emit_relations=1, emit_origin=1, ddl_every=rng.choice([2, 3, 7]), type_msg_pct=10,
                               keepalive_every=97, name="ddl_dense")
        elif mode == "ddl_fixed":   # fixed-width tables only, a DDL message -> Relation now and then, default flags: batches WITHOUT a control frame behind one that had
            # some take the plan kernel behind their control pass (standard_path, round 6); Updates by key / Deletes hand those back to the generic kernel
            w = synth.Workload([synth.table_fixed(), synth.table_fixed(16390, "bench_fixed_b")], 0xE7F0000 + seed * 1000 + rounds, rows_per_txn=rng.choice([5, 12, 40]),
                               mix=rng.choice([(100, 0, 0), (90, 10, 0), (60, 30, 10)]), upd_key=rng.choice([0, 0, 10]), emit_relations=1, emit_origin=1,
                               ddl_every=rng.choice([3, 7, 40, 200]), keepalive_every=rng.choice([0, 997]), name="ddl_fixed")
        else:
            w = synth.cfg5(seed=0xE7B0000 + seed * 1000 + rounds)
        buf, offs = w.fill((rng.choice([2, 3, 4]) << 20) if mode == "cfg5" else (rng.choice([1, 2]) << 20))
        pieces = _cut(buf, offs, rng.randrange(80, 160), seed=seed * 104729 + rounds)
        o, d = oracle.Oracle(), Decoder(0)
        w.register(o, ready=not w.cfg.emit_relations)
        w.register(d, ready=not w.cfg.emit_relations)
        dev = DevBufs(pieces)
        window = rng.choice([1, 2, 3, 6, 12, 20])
        inflight, done = [], 0
        for k, (p, nbytes, po, nf) in enumerate(dev.items):
            inflight.append(d.decode_device(p, nbytes, po, nf, FLAGS_DEFAULT))
            while len(inflight) - done > window or (k == len(dev.items) - 1 and done < len(inflight)):
                b = inflight[done]
                rb = o.decode(*pieces[done])
                rc = b.sync()
                diff = [] if (rc != 0 or rb.err_code != 0) else rb.host_batch().diff(b.host())
                if rc != 0 or rb.err_code != 0 or diff:
                    bad += 1
                    print("MISMATCH cfg5 seed", seed, "round", rounds, "batch", done, "of", len(pieces), "window", window, "rc", rc, b.error.description if b.error else "", rb.err_code, diff[:4], flush=True)
                b.close()
                done += 1
                batches += 1
        paths = d.debug_paths()
        d.close()
        rounds += 1
        print("round", rounds, "window", window, "batches", len(pieces), paths, flush=True)
    print(f"async long fuzz ({mode}, default flags): {rounds} chains, {batches} batches checked, {bad} problems, seeds {seed}..{seed + rounds - 1}")
    sys.exit(1 if bad else 0)
t_end = time.time() + seconds
rounds = batches = bad = spliced_total = 0
while time.time() < t_end or rounds == 0:
    rng = random.Random(seed + rounds)
    w = synth.cfg2(seed=0xE7A0000 + seed * 1000 + rounds)
    n = rng.randrange(80, 200)
    buf, offs = w.fill(rng.choice([2, 3, 5]) << 20)
    pieces = _cut(buf, offs, n, seed=seed * 7919 + rounds)
    spliced = sorted(rng.sample(range(3, len(pieces)), rng.choice([1, 1, 2, 3, 5])))
    for k in spliced:
        b1, o1 = pieces[k]
        at = None
        for cand in range(1, len(o1) - 1):
            if b1[int(o1[cand]) + 30] == ord("I") and b1[int(o1[cand - 1]) + 30] == ord("I"):
                at = cand
                break
        if at is None:
            continue
        rel = int.from_bytes(b1[int(o1[at]) + 31:int(o1[at]) + 35].tobytes(), "big")
        lsn = int.from_bytes(b1[int(o1[at]) + 6:int(o1[at]) + 14].tobytes(), "big")
        upd = W.frame(W.xlog(lsn - 1, W.update(rel, ["5", "6", "7", "8", "9"], key=["4"])))
        cut = int(o1[at])
        pieces[k] = (np.concatenate([b1[:cut], np.frombuffer(upd, dtype=np.uint8), b1[cut:]]),
                     np.concatenate([o1[:at + 1], o1[at:] + len(upd)]).astype(np.uint32))
    # Deletes by key in a third of the batches (round 6, last session: the fixed-width plan decodes them itself — dense and full-width key tuples): an
    # Insert replaced by the Delete of its row, same LSN
    for k in range(len(pieces)):
        if k in spliced or rng.random() > 0.33:
            continue
        b1, o1 = pieces[k]
        ins = [f for f in range(len(o1) - 1) if b1[int(o1[f]) + 30] == ord("I")]
        for f in sorted(rng.sample(ins, min(len(ins), rng.choice([1, 1, 2, 5]))), reverse=True):
            fr = bytes(b1[int(o1[f]):int(o1[f + 1])])
            cl = int.from_bytes(fr[39:43], "big")
            cells = fr[38:43 + cl] + (b"n" * 4 if rng.random() < 0.5 else b"")
            body = b"D" + fr[31:35] + b"K" + (5 if len(cells) > 5 + cl else 1).to_bytes(2, "big") + cells
            nf = b"d" + (4 + 25 + len(body)).to_bytes(4, "big") + fr[5:30] + body
            b1 = np.concatenate([b1[:int(o1[f])], np.frombuffer(nf, dtype=np.uint8), b1[int(o1[f + 1]):]])
            o1 = o1.astype(np.int64).copy()
            o1[f + 1:] += len(nf) - len(fr)
            o1 = o1.astype(np.uint32)
        pieces[k] = (b1, o1)
    broken = []
    if mode == "errors":   # a malformed integer in a few batches: the batch ends there with the reference's error, the context keeps the state before
        for k in sorted(rng.sample(range(3, len(pieces)), rng.choice([1, 2, 3]))):   # the failing frame, and the chain goes on from it (both sides alike)
            b2, o2 = pieces[k]
            for cand in range(len(o2) - 1):
                if b2[int(o2[cand]) + 30] == ord("I"):
                    b2[int(o2[cand]) + 43 + 3] = ord("x")
                    broken.append(k)
                    break
    o, d = oracle.Oracle(), Decoder(0)
    w.register(o)
    w.register(d)
    dev = DevBufs(pieces)
    window = rng.choice([1, 2, 3, 6, 12, 20, 30])
    inflight, done = [], 0
    for k, (p, nbytes, po, nf) in enumerate(dev.items):
        sidecar = mode != "nosidecar" and not (mode == "mixed" and rng.random() < 0.5)   # without one: the boundary scan of the batch runs ahead, its decode is enqueued by the next call
        inflight.append(d.decode_device(p, nbytes, po if sidecar else None, nf if sidecar else 0, FLAGS))
        while len(inflight) - done > window or (k == len(dev.items) - 1 and done < len(inflight)):
            b = inflight[done]
            rb = o.decode(*pieces[done])
            rc = b.sync()
            if mode == "errors":
                e = b.error
                got = (e.code, e.frame_index) if e else (0, -1)
                same_err = got == (rb.err_code, rb.err_frame)
                diff = rb.host_batch().diff(b.host()) if same_err else ["error %s, oracle (%d, %d)" % (got, rb.err_code, rb.err_frame)]
                rc_bad = False
            else:
                diff = [] if (rc != 0 or rb.err_code != 0) else rb.host_batch().diff(b.host())
                rc_bad = rc != 0 or rb.err_code != 0
            if rc_bad or diff:
                bad += 1
                print("MISMATCH seed", seed, "round", rounds, "batch", done, "of", len(pieces), "window", window, "spliced", spliced, "broken", broken, "rc", rc, rb.err_code, diff[:4], flush=True)
            b.close()
            done += 1
            batches += 1
    paths = d.debug_paths()
    if mode != "errors" and os.environ.get("ETLG_PLAN_DELETES", "1") != "0" and (paths["chain_rerun"] > (window + 1) * len(spliced) or paths["redone"]):
        bad += 1
        print("CHAIN DID NOT HEAL seed", seed, "round", rounds, "window", window, "spliced", spliced, paths, flush=True)
    spliced_total += len(spliced)
    d.close()
    rounds += 1
print(f"async long fuzz: {rounds} chains, {batches} batches checked, {spliced_total} batches with an UPDATE, {bad} problems, seeds {seed}..{seed + rounds - 1}")
sys.exit(1 if bad else 0)
