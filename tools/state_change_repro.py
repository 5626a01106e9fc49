"""Diagnostics for a round of tools/async_fuzz.py whose table-state change goes wrong on the MI355X: python tools/state_change_repro.py seed"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from etl_amd import abi, synth
from etl_amd.decoder import Decoder
from oracle import oracle
from tests.test_gpu_async import DevBufs, _cut
seed = int(sys.argv[1])
rng = random.Random(seed)
mk = rng.choice([synth.cfg2, synth.cfg2, synth.cfg3, synth.cfg5]); w = mk()
for _ in range(rng.randrange(3)): w.fill(rng.choice([1 << 16, 1 << 18]))
buf, offs = w.fill(rng.choice([1 << 19, 1 << 20, 3 << 20])); buf = buf.copy()
damage = rng.random() < 0.35
if damage:
    for _ in range(rng.randrange(1, 4)):
        f = rng.randrange(len(offs) - 1); lo, hi = int(offs[f]), int(offs[f + 1])
        pos = rng.randrange(lo + 30, hi) if hi - lo > 31 else lo
        buf[pos] = rng.choice([0, 0x2D, 0x41, 0xFF, 0x6E, 0x75, 0x74])
nparts = rng.randrange(3, 41)
if nparts >= len(offs) - 1: nparts = 2
pieces = _cut(buf, offs, nparts, seed=seed)
no_ctrl = mk is not synth.cfg5 and rng.random() < 0.6
sidecar = rng.random() < 0.75
window = rng.choice([1, 2, 3, 8, 24])
host_in = rng.random() < 0.25; side_calls = rng.random() < 0.3
state_changes = {}
if rng.random() < 0.3 and not no_ctrl:
    begins = [int.from_bytes(bytes(buf[int(x) + 31:int(x) + 39]), "big") for x in offs[:-1] if buf[int(x) + 5] == ord("w") and buf[int(x) + 30] == ord("B")]
    for _ in range(rng.randrange(1, 4)):
        at = rng.randrange(1, len(pieces)); t = rng.choice(w.tables)
        kind = rng.choice([abi.TS_READY, abi.TS_SYNC_DONE, abi.TS_SYNC_DONE, abi.TS_OTHER, abi.TS_ABSENT])
        lsn = (rng.choice(begins) + rng.choice([-1, 0, 1])) if begins and kind == abi.TS_SYNC_DONE else 0
        state_changes.setdefault(at, []).append((t["rel_id"], kind, max(lsn, 0)))
print(mk.__name__, "parts", nparts, "window", window, "sidecar", sidecar, "no_ctrl", no_ctrl, "damage", damage, "host_in", host_in, "state_changes", state_changes)
flags = abi.F_INPUT_ON_DEVICE | abi.F_OUTPUT_ON_DEVICE | abi.F_ASYNC
for trial in range(3):
    o, d = oracle.Oracle(), Decoder(0)
    w2 = mk(); w2.register(o, ready=not w2.cfg.emit_relations); w2.register(d, ready=not w2.cfg.emit_relations)
    dev = DevBufs(pieces)
    for k, (p, n, po, nf) in enumerate(dev.items):
        if k in state_changes:
            for rel, kind, lsn in state_changes[k]:
                o.table_state(rel, kind, lsn); d.table_state(rel, kind, lsn)
        b = d.decode_device(p, n, po, nf, flags)          # one batch at a time: window 1
        rb = o.decode(*pieces[k]); rc = b.sync()
        hb, ob = b.host(), rb.host_batch()
        df = ob.diff(hb)
        if df or k in state_changes:
            tags = bytes(pieces[k][0][int(x) + 30] for x in pieces[k][1][:-1])
            print(" trial", trial, "batch", k, "changed" if k in state_changes else "", "frames", len(tags), "first tags", tags[:12], "oracle events", ob.n_events, "device", hb.n_events, "paths", d.debug_paths(), "diff", df[:2])
            if df and ob.n_events and hb.n_events:
                print("    oracle commit", ob.commit_lsn[:4], "device", hb.commit_lsn[:4])
    d.close()
