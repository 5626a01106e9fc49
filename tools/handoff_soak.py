"""A long-lived context against fresh ones (emulator or GPU): python tools/handoff_soak.py [seconds=60] [seed=1]
Every other harness of the hand-off calls makes a context per batch, so the pools a context keeps across batches — output arenas,
column / row buffers given back by etlg_columns_free / etlg_rowbinary_free, pinned result blocks, side-input sets, the result ring —
are only ever used once. Here ONE context decodes hundreds of batches of the cfg3 schema (inserts, updates, deletes; TEXT / NUMERIC /
timestamptz / uuid), synchronously or ASYNC, and after each a random subset of { Arrow columns, RowBinary rows, BigQuery rows, size
hints } is built from the device-resident arena — results kept alive for a random number of rounds and freed out of order, batches freed
before or after their results. Each result must equal, byte for byte, what a FRESH context makes of the same batch (the fresh path is
what tests/test_gpu_columns.py / _rowbinary / _protobuf / _size_hints pin to the oracle)."""
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
from oracle import size_hint as SH

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
SM = abi.SizeModel()
for k, v in SH.MODEL.items():
    setattr(SM, k, v)


def snapshot(kind, res):
    """The bytes of a result, host side."""
    if kind == "columns":
        out = [res.n_rows, res.row_event().tobytes()]
        for i in range(int(res.view.n_cols)):
            ha = res.host_arrays(i)
            if ha is None:
                out.append(None)
                continue
            v, dfr, vals, offs = ha
            out.append((int(res.column(i).arrow_kind), v.tobytes(), dfr.tobytes(), vals.tobytes(), None if offs is None else offs.tobytes()))
        return out
    if kind in ("rowbinary", "protobuf"):
        return [res.status, res.n_rows, res.bytes().tobytes(), res.row_offsets().tobytes(), res.row_event().tobytes()]
    return res.tobytes()   # size hints: a numpy array


def build(kind, b, nullable):
    if kind == "columns":
        return b.columns(0, kinds=("I", "U"))
    if kind == "rowbinary":
        return b.rowbinary(0, nullable + [0, 0], abi.CH_REPLACING_MERGE_TREE)
    if kind == "protobuf":
        return b.protobuf(0)
    return b.size_hints(SM)


w = synth.cfg3(seed=0xE7C0000 + seed)
pool = [w.fill(rng.choice([64, 200, 700, 1500]) << 10) for _ in range(12)]
long_ctx = Decoder(0)
w.register(long_ctx)
c0 = None
t_end = time.time() + seconds
rounds = checks = bad = 0
alive = []   # (rounds to live, result object or None, batch or None)
nullable = None
while time.time() < t_end or rounds == 0:
    buf, offs = pool[rng.randrange(len(pool))]
    long_ctx.reset_stream_state()
    use_async = rng.random() < 0.5
    b = long_ctx.decode(buf, offs, flags=abi.F_OUTPUT_ON_DEVICE | (abi.F_ASYNC if use_async else 0)) if not use_async else None
    if use_async:   # (host input + ASYNC: staged upload, joins the chain)
        b = long_ctx.decode(buf, offs, flags=abi.F_OUTPUT_ON_DEVICE | abi.F_ASYNC)
        if rng.random() < 0.7:
            assert b.sync() == 0
    fresh = Decoder(0)
    w.register(fresh)
    fb = fresh.decode(buf, offs, flags=abi.F_OUTPUT_ON_DEVICE)
    assert fb.rc == 0
    if nullable is None:
        cc = fb.columns(0)
        nullable = [1 if cc.column(i).nullable else 0 for i in range(int(cc.view.n_cols))]
        cc.close()
    kinds = rng.sample(["columns", "rowbinary", "protobuf", "hints"], rng.randrange(1, 5))
    for kind in kinds:
        try:
            if kind == "hints" and use_async:
                assert b.sync() == 0            # (the wrapper sizes its output array from the batch's event count)
            got = build(kind, b, nullable)
            want = build(kind, fresh and fb, nullable)
        except Exception as e:
            bad += 1
            print("ERROR round", rounds, kind, repr(e)[:200], flush=True)
            continue
        gs, ws = snapshot(kind, got), snapshot(kind, want)
        checks += 1
        if gs != ws:
            bad += 1
            print("MISMATCH round", rounds, kind, "async" if use_async else "sync", flush=True)
        if kind != "hints":
            want.close()
            if rng.random() < 0.5:
                alive.append([rng.randrange(1, 6), got, None])   # kept alive: its buffers must not be handed to a later result
            else:
                got.close()
    fb.close()
    fresh.close()
    if rng.random() < 0.5:
        alive.append([rng.randrange(1, 4), None, b])             # the batch outlives the round (its arena stays out of the pool)
    else:
        b.close()
    nxt = []
    for ttl, res, bt in alive:
        if ttl <= 1:
            (res or bt).close()
        else:
            nxt.append([ttl - 1, res, bt])
    rng.shuffle(nxt)
    alive = nxt
    rounds += 1
for _ttl, res, bt in alive:
    (res or bt).close()
long_ctx.close()
print(f"hand-off soak: {rounds} batches on one context, {checks} results compared with a fresh context's, {bad} problems (seed {seed})")
sys.exit(1 if bad else 0)
