"""Randomised check of the multi-GPU recipe (etl_amd/shard.py: commit-aligned cuts, control streams broadcast and replayed, per-shard
decode, concatenation) against ONE context's decode of the same stream, on the real library or the emulator build:
python tools/shard_fuzz.py [seconds=60] [seed=1]
Every round: a cfg3 or cfg5 stream that starts at a random point, 256 KiB - 3 MiB, cut into 1-12 shards; the concatenation of the
shards' arenas must equal the single decode byte for byte, and the control stream the DEVICE extracts for every shard
(etlg_control_stream) must equal the host reference's."""
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from etl_amd import shard, synth
from etl_amd.decoder import Decoder
from etl_amd.view import HostBatch

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + seconds
rounds = bad = 0
seed = seed0
while time.time() < t_end or rounds == 0:
    rng = random.Random(seed)
    mk = rng.choice([synth.cfg3, synth.cfg5, synth.cfg5])
    w = mk()
    skip = [rng.choice([1 << 16, 1 << 18, 1 << 20]) for _ in range(rng.randrange(3))]
    nbytes = rng.choice([1 << 18, 1 << 20, 3 << 20])
    n = rng.randrange(1, 13)
    # the single decode sees the whole stream from its start (so do the shards: the skipped part is one leading piece of both)
    bufs = [w.fill(k) for k in skip]
    buf, offs = w.fill(nbytes)
    made = []

    def mkctx():
        d = Decoder(0)
        w2 = mk()
        w2.register(d, ready=not w2.cfg.emit_relations)
        for pb, po in bufs:            # every context first walks the skipped part (its Relation frames, its transaction state)
            r = d.decode(pb, po)
            assert r.rc == 0, r.error
        made.append(d)
        return d
    what = f"seed {seed} {mk.__name__} skip {skip} bytes {nbytes} shards {n}"
    try:
        one = mkctx()
        ref = one.decode(buf, offs)
        assert ref.rc == 0, ref.error
        parts = shard.decode_sharded(mkctx, buf, offs, n)
        got = HostBatch.concat([b.host() for _, b in parts])
        diff = ref.host().diff(got)
        if diff:
            bad += 1
            print("MISMATCH", what, diff[:4], flush=True)
        for f0, f1 in shard.plan_shards(buf, offs, n):
            if f1 == f0:
                continue
            b, oo = shard.slice_shard(buf, offs, f0, f1)
            b, oo = np.ascontiguousarray(b), np.ascontiguousarray(oo, dtype=np.uint32)
            want_b, want_o = shard.control_stream(b, oo)
            got_b, got_o, last = one.control_stream(b.ctypes.data, len(b), oo.ctypes.data, len(oo) - 1, on_device=False)
            if not (np.array_equal(got_o, want_o) and np.array_equal(got_b, want_b)):
                bad += 1
                print("MISMATCH control stream", what, "shard", f0, f1, flush=True)
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("ERROR", what, repr(e)[:300], flush=True)
    for d in made:
        d.close()
    rounds += 1
    seed += 1
print(f"shard fuzz: {rounds} rounds, {bad} problems, seeds {seed0}..{seed - 1}")
sys.exit(1 if bad else 0)
