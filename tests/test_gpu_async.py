"""ETLG_F_ASYNC pipelining: batches enqueued back to back, before any of them is synced, must start from the transaction
state their predecessor LEFT ON THE DEVICE (DecParams.carry), not from what the host knew when it enqueued them — the
apply loop carries remote_final_lsn / the next ordinal from message to message (crates/etl/src/replication/apply.rs:942-963,
2284-2292). The batches here are cut at arbitrary frame boundaries, so transactions span them.

Also: a batch that cannot be produced by its first kernel (an error; a frame the fixed-width plan does not cover) poisons the
batches queued behind it; they are decoded again, in order, when they are synced, and must still match the oracle."""
import os

import numpy as np
import pytest

from etl_amd import abi, synth

pytestmark = pytest.mark.gpu

FLAGS = abi.F_INPUT_ON_DEVICE | abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL | abi.F_ASYNC


class DevBufs:
    """Device copies of (bytes, offsets). On the CPU emulator build "device memory" is host memory."""

    def __init__(self, pieces):
        self.keep = []
        self.items = []
        emu = os.environ.get("ETLG_SIMT_RUN") == "1"
        if not emu:
            import torch
        for buf, offs in pieces:
            o32 = np.ascontiguousarray(offs.astype(np.uint32))
            if emu:
                b = np.ascontiguousarray(buf)
                self.keep += [b, o32]
                self.items.append((b.ctypes.data, len(b), o32.ctypes.data, len(o32) - 1))
            else:
                tb = torch.from_numpy(np.ascontiguousarray(buf).copy()).cuda()
                to = torch.from_numpy(o32.view(np.int32).copy()).cuda()
                self.keep += [tb, to]
                self.items.append((tb.data_ptr(), tb.numel(), to.data_ptr(), len(o32) - 1))
        if not emu:
            torch.cuda.synchronize()


def _cut(buf, offs, nparts, seed):
    """Cuts one framed stream into `nparts` consecutive batches at pseudo-random FRAME boundaries (not commit aligned)."""
    rng = np.random.default_rng(seed)
    nf = len(offs) - 1
    cuts = sorted(set(int(x) for x in rng.integers(1, nf, nparts - 1)))
    edges = [0] + cuts + [nf]
    out = []
    for a, b in zip(edges[:-1], edges[1:]):
        o = offs[a:b + 1].astype(np.int64)
        out.append((buf[int(o[0]):int(o[-1])].copy(), (o - o[0]).astype(np.uint32)))
    return out


def _run_chain(w, pieces, expect_path=None, sidecar=True):
    from etl_amd.decoder import Decoder
    from oracle import oracle
    o, d = oracle.Oracle(), Decoder(0)
    w.register(o)
    w.register(d)
    dev = DevBufs(pieces)
    inflight = [d.decode_device(p, n, po if sidecar else None, nf if sidecar else 0, FLAGS) for (p, n, po, nf) in dev.items]   # nothing synced yet
    res = []
    for (buf, offs), b in zip(pieces, inflight):
        rb = o.decode(buf, offs)
        rc = b.sync()
        res.append((rb, rc, b))
    paths = d.debug_paths()
    paths["overlapped"] = d.debug_overlapped()
    for i, (rb, rc, b) in enumerate(res):
        assert (rb.err_code != 0) == (rc != 0), f"batch {i}: oracle error {rb.err_code} vs rc {rc} ({b.error})"
        if rb.err_code:
            assert (b.error.code, b.error.frame_index) == (rb.err_code, rb.err_frame), f"batch {i}"
        diff = rb.host_batch().diff(b.host())
        assert not diff, f"batch {i}: {diff[:6]}"
        b.close()
    d.close()
    return paths


def async_fuzz_round(seed):
    """One round of tools/async_fuzz.py: a synthetic stream (cfg2 / cfg3 / cfg5, sometimes started inside the stream, sometimes with a
    few damaged bytes), cut into 2-40 batches at arbitrary frame boundaries, enqueued ETLG_F_ASYNC with a random window of batches in
    flight, with or without the offsets sidecar and the caller's no-control assertion, synced in issue order; every batch against the
    oracle. Returns (batches checked, list of mismatch descriptions)."""
    import random
    from etl_amd.decoder import Decoder
    from oracle import oracle
    rng = random.Random(seed)
    mk = rng.choice([synth.cfg2, synth.cfg2, synth.cfg3, synth.cfg5])
    w = mk()
    for _ in range(rng.randrange(3)):
        w.fill(rng.choice([1 << 16, 1 << 18]))          # start somewhere inside the stream
    buf, offs = w.fill(rng.choice([1 << 19, 1 << 20, 3 << 20]))
    buf = buf.copy()
    damage = rng.random() < 0.35
    if damage:
        for _ in range(rng.randrange(1, 4)):
            f = rng.randrange(len(offs) - 1)
            lo, hi = int(offs[f]), int(offs[f + 1])
            pos = rng.randrange(lo + 30, hi) if hi - lo > 31 else lo
            buf[pos] = rng.choice([0, 0x2D, 0x41, 0xFF, 0x6E, 0x75, 0x74])
    nparts = rng.randrange(3, 41)
    if nparts >= len(offs) - 1:
        nparts = 2
    pieces = _cut(buf, offs, nparts, seed=seed)
    no_ctrl = mk is not synth.cfg5 and rng.random() < 0.6
    sidecar = rng.random() < 0.75
    window = rng.choice([1, 2, 3, 8, 24])
    # (drawn after everything above, so that the rounds of earlier seeds stay what they were)
    host_in = rng.random() < 0.25          # host buffers: with a sidecar they are staged on the copy stream, without one decoded synchronously
    side_calls = rng.random() < 0.3        # other calls of the ABI between the batches of the chain
    flags = (0 if host_in else abi.F_INPUT_ON_DEVICE) | abi.F_OUTPUT_ON_DEVICE | abi.F_ASYNC | (abi.F_NO_CONTROL if no_ctrl else 0)
    # library knobs read when the context is created (drawn from a generator of their own: the rounds stay what they were): the plan's
    # sidecar pre-pass on / off / in front of two tiles per wave, the order of the pipelined control path
    krng = random.Random(seed * 7919 + 13)
    knobs = {"ETLG_PLAN_PRE": krng.choice(["1", "1", "0", "2"]), "ETLG_CTL_HOLD": krng.choice(["1", "1", "0"])}
    saved_knobs = {k: os.environ.get(k) for k in knobs}
    os.environ.update(knobs)
    try:
        o, d = oracle.Oracle(), Decoder(0)
    finally:
        for k, v in saved_knobs.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    ready = not w.cfg.emit_relations
    w.register(o, ready=ready)
    w.register(d, ready=ready)
    dev = DevBufs(pieces)
    if host_in:
        keep = [(np.ascontiguousarray(pb), np.ascontiguousarray(po.astype(np.uint32))) for pb, po in pieces]
        dev.keep.append(keep)
        dev.items = [(pb.ctypes.data, len(pb), po.ctypes.data, len(po) - 1) for pb, po in keep]
    what = f"seed {seed} {mk.__name__} parts {nparts} window {window} sidecar {sidecar} no_ctrl {no_ctrl} damaged {damage} host_in {host_in} side_calls {side_calls}"
    if os.environ.get("ETLG_FUZZ_VERBOSE"):
        print(what, flush=True)
    st = {"done": 0, "stop": False, "bad": []}
    inflight = []

    def collect():
        k = st["done"]
        b = inflight[k]
        rb = o.decode(*pieces[k])
        rc = b.sync()
        ok = (rb.err_code != 0) == (rc != 0)
        if ok and rb.err_code:
            ok = (b.error.code, b.error.frame_index) == (rb.err_code, rb.err_frame)
        diff = rb.host_batch().diff(b.host()) if ok else ["error: oracle %s vs rc %s %s" % (rb.err_code, rc, b.error)]
        if diff:
            st["bad"].append(f"{what} batch {k}: {diff[:3]}")
            st["stop"] = True
        if rb.err_code:
            st["stop"] = True      # the reference's apply loop exits on a decode error (apply.rs:2475-2481): the chain ends here
        b.close()
        st["done"] += 1

    # table-state changes between batches (the apply loop sees its state store change between messages: a table finishes its copy —
    # SyncDone(lsn): owned from that LSN on, decided per transaction, the kernels' seq_lookback mode —, errors out, comes back): the
    # library finishes the chain before it takes the change, so the round does the same with the oracle
    state_changes = {}
    if rng.random() < 0.3 and not no_ctrl:
        begins = [int.from_bytes(bytes(buf[int(x) + 31:int(x) + 39]), "big") for x in offs[:-1] if buf[int(x) + 5] == ord("w") and buf[int(x) + 30] == ord("B")]
        for _ in range(rng.randrange(1, 4)):
            at = rng.randrange(1, len(pieces))
            t = rng.choice(w.tables)
            kind = rng.choice([abi.TS_READY, abi.TS_SYNC_DONE, abi.TS_SYNC_DONE, abi.TS_OTHER, abi.TS_ABSENT])
            lsn = (rng.choice(begins) + rng.choice([-1, 0, 1])) if begins and kind == abi.TS_SYNC_DONE else 0
            state_changes.setdefault(at, []).append((t["rel_id"], kind, max(lsn, 0)))
        what += f" state_changes {sorted(state_changes)}"
    for kk, (p, n, po, nf) in enumerate(dev.items):
        if st["stop"]:
            break
        if kk in state_changes:
            while not st["stop"] and st["done"] < len(inflight):
                collect()
            if st["stop"]:
                break
            for rel, kind, lsn in state_changes[kk]:
                o.table_state(rel, kind, lsn)
                d.table_state(rel, kind, lsn)
        enqueue = d.decode_host_ptr if host_in else d.decode_device
        inflight.append(enqueue(p, n, po if sidecar else None, nf if sidecar else 0, flags))
        if side_calls and not damage and rng.random() < 0.4:   # a boundary scan / a tag pass of some piece in the middle of the chain: right answers, chain undisturbed
            j = rng.randrange(len(pieces))
            pb, po = pieces[j]
            if rng.random() < 0.5:
                got = d.scan_boundaries(pb)
                if not np.array_equal(got, po.astype(np.uint32)):
                    st["bad"].append(f"{what}: scan_boundaries of piece {j} between batches {kk} and {kk + 1}")
            elif len(po) > 1:
                tags = d.frame_tags(pb, po)
                want = np.array([pb[int(x) + 5] if pb[int(x) + 5] == ord("k") else pb[int(x) + 30] for x in po[:-1]], dtype=np.uint8)   # keepalives: 'k'
                if not np.array_equal(np.asarray(tags, dtype=np.uint8), want):
                    st["bad"].append(f"{what}: frame_tags of piece {j} between batches {kk} and {kk + 1}")
        while not st["stop"] and len(inflight) - st["done"] >= window:
            collect()
    checked = 0
    while st["done"] < len(inflight):
        if st["stop"]:
            inflight[st["done"]].sync()
            inflight[st["done"]].close()
            st["done"] += 1
        else:
            collect()
    checked = st["done"]
    d.close()
    return checked, st["bad"]


@pytest.mark.parametrize("seed", [61, 106, 226, 261, 95274, 95621, 7, 19, 42])
def test_async_fuzz_rounds(seed):
    """Rounds of tools/async_fuzz.py that found defects in round 4, plus a few more. 61 / 106: optimistic batches queued behind the
    batch that carries the stream's first Relation frames listed no schema slots in their views. 226: the control pre-pass (and host
    control plane) of a later batch ran ahead of optimistic batches that were still pending and had to be decoded again — against the
    table's NEXT schema. 261: a batch whose boundary scan was deferred took the control path from the state of the last synced batch
    while the batches in between were pending. 95274 / 95621 (MI355X only — the emulator cannot show it): a table goes SyncDone between
    two batches of a chain; k_cells then runs the transaction look-back in its first phase, and its wave read the LDS slots lane 0 had
    just written with no barrier in between — the compiler is free to order that either way (stale transaction state: rows of skipped
    transactions were emitted). The values now stay in registers (lookback.hip.h TxnStart)."""
    checked, bad = async_fuzz_round(seed)
    assert checked >= 1 and not bad, bad


@pytest.mark.parametrize("mk,nbytes", [(synth.cfg2, 3 << 20), (synth.cfg3, 3 << 20)])
def test_async_chain_carries_transaction_state(mk, nbytes):
    w = mk()
    buf, offs = w.fill(nbytes)
    pieces = _cut(buf, offs, 7, seed=11)
    paths = _run_chain(w, pieces)
    assert paths["redone"] == 0 and paths["chain_rerun"] == 0 and paths["plan_redone"] == 0, paths
    if mk is synth.cfg2:
        assert paths["plan"] == 7, paths
        # consecutive plan batches run side by side on two streams; the transactions span the cuts, so the first tiles of every
        # batch had to wait for the state their predecessor left (plan.hip: plan_late_carry)
        assert paths["overlapped"] >= 4, paths


def test_async_chain_long_two_streams_and_ring_laps():
    """70 cfg2 batches cut at arbitrary frames, up to 20 in flight: the result ring (32 blocks) laps twice, the look-back
    buffers rotate, and every batch but the first of a lap runs beside its predecessor on the other decode stream."""
    from etl_amd.decoder import Decoder
    from oracle import oracle
    w = synth.cfg2()
    buf, offs = w.fill(6 << 20)
    pieces = _cut(buf, offs, 70, seed=23)
    o, d = oracle.Oracle(), Decoder(0)
    w.register(o)
    w.register(d)
    dev = DevBufs(pieces)
    inflight = []
    done = 0
    for k, (p, n, po, nf) in enumerate(dev.items):
        inflight.append(d.decode_device(p, n, po, nf, FLAGS))
        while len(inflight) - done > 20 or (k == len(dev.items) - 1 and done < len(inflight)):
            b = inflight[done]
            rb = o.decode(*pieces[done])
            assert b.sync() == 0 and rb.err_code == 0
            diff = rb.host_batch().diff(b.host())
            assert not diff, f"batch {done}: {diff[:6]}"
            b.close()
            done += 1
    paths = d.debug_paths()
    assert paths["plan"] == 70 and paths["chain_rerun"] == 0 and paths["plan_redone"] == 0, paths
    assert d.debug_overlapped() >= 60, d.debug_overlapped()
    d.close()


def test_async_chain_survives_an_error_in_the_middle():
    """Batch 2 holds a malformed integer: its error must be the oracle's, and the batches queued behind it — which the device
    refused to run from a failed predecessor — are decoded when they are synced, from the state the failed batch left."""
    w = synth.cfg2()
    buf, offs = w.fill(2 << 20)
    pieces = _cut(buf, offs, 6, seed=5)
    b2, o2 = pieces[2]
    fr = int(o2[len(o2) // 2])          # some frame in the middle of batch 2
    k = fr
    while b2[k + 30] != ord("I"):       # find an Insert at or after it
        k = int(o2[np.searchsorted(o2, k, side="right")])
    b2[k + 43 + 3] = ord("x")           # a letter inside the first int4 text
    paths = _run_chain(w, pieces)
    assert paths["chain_rerun"] >= 1, paths


def test_async_chain_when_the_plan_does_not_cover_a_batch():
    """An UPDATE that carries a key image (an Update without an old image is the plan's own since round 6) in a stream of fixed-width
    inserts: k_plan gives the batch up, the generic kernel decodes it, and the batches queued behind it run again from the right state."""
    from tests import pgwire as W
    w = synth.cfg2()
    buf, offs = w.fill(1 << 20)
    pieces = _cut(buf, offs, 5, seed=3)
    # splice an Update of the same table in front of a frame of batch 1 that sits inside a transaction
    b1, o1 = pieces[1]
    nf = len(o1) - 1
    at = nf // 2
    while b1[int(o1[at]) + 30] != ord("I") or b1[int(o1[at - 1]) + 30] != ord("I"):
        at += 1
    rel = int.from_bytes(b1[int(o1[at]) + 31:int(o1[at]) + 35].tobytes(), "big")
    lsn = int.from_bytes(b1[int(o1[at]) + 6:int(o1[at]) + 14].tobytes(), "big")
    upd = W.frame(W.xlog(lsn - 1, W.update(rel, ["5", "6", "7", "8", "9"], key=["4"])))
    cut = int(o1[at])
    nb = np.concatenate([b1[:cut], np.frombuffer(upd, dtype=np.uint8), b1[cut:]])
    no = np.concatenate([o1[:at + 1], o1[at:] + len(upd)]).astype(np.uint32)
    pieces[1] = (nb, no)
    paths = _run_chain(w, pieces)
    assert paths["plan_redone"] >= 1 and paths["redone"] == 0, paths


@pytest.mark.parametrize("spare", [True, False])
def test_async_chain_second_attempts_across_a_ring_lap(spare, monkeypatch):
    """70 cfg2 batches, six in flight; batch 30 holds an UPDATE with a key image, which the fixed-width plan does not cover. It is decoded again when it is
    synced — and with it the batches queued behind it — AFTER batch 32 (result block 0) has sent out the ring's re-initialisation for the
    next lap: what the second attempts leave in blocks 30 and 31 must not meet batches 62 and 63 (payload shards are added to, give-up
    and error words or-ed / min-ed into, carry_ready polled by the batch that runs beside). Every batch of the chain against the oracle.
    `spare` (the default since round 6's last session): the generic kernel's second attempt at batch 30 leaves the carried transaction
    state the plan had published before it gave up, so the batches behind it STAND — one second attempt, no chain to heal; with
    ETLG_CHAIN_SPARE=0 they are decoded again as before."""
    from etl_amd.decoder import Decoder
    monkeypatch.setenv("ETLG_CHAIN_SPARE", "1" if spare else "0")
    from oracle import oracle
    from tests import pgwire as W
    w = synth.cfg2()
    buf, offs = w.fill(6 << 20)
    pieces = _cut(buf, offs, 70, seed=29)
    b1, o1 = pieces[30]
    at = (len(o1) - 1) // 2
    while b1[int(o1[at]) + 30] != ord("I") or b1[int(o1[at - 1]) + 30] != ord("I"):
        at += 1
    rel = int.from_bytes(b1[int(o1[at]) + 31:int(o1[at]) + 35].tobytes(), "big")
    lsn = int.from_bytes(b1[int(o1[at]) + 6:int(o1[at]) + 14].tobytes(), "big")
    upd = W.frame(W.xlog(lsn - 1, W.update(rel, ["5", "6", "7", "8", "9"], key=["4"])))
    cut = int(o1[at])
    pieces[30] = (np.concatenate([b1[:cut], np.frombuffer(upd, dtype=np.uint8), b1[cut:]]),
                  np.concatenate([o1[:at + 1], o1[at:] + len(upd)]).astype(np.uint32))
    o, d = oracle.Oracle(), Decoder(0)
    w.register(o)
    w.register(d)
    dev = DevBufs(pieces)
    inflight = []
    done = 0
    for k, (p, n, po, nf) in enumerate(dev.items):
        inflight.append(d.decode_device(p, n, po, nf, FLAGS))
        while len(inflight) - done > 6 or (k == len(dev.items) - 1 and done < len(inflight)):
            b = inflight[done]
            rb = o.decode(*pieces[done])
            assert b.sync() == 0 and rb.err_code == 0, (done, b.error)
            diff = rb.host_batch().diff(b.host())
            assert not diff, f"batch {done}: {diff[:6]}"
            b.close()
            done += 1
    paths = d.debug_paths()
    if spare:
        assert paths["plan_redone"] == 1 and paths["chain_rerun"] == 0 and d.debug_chains_spared() == 1, paths
        assert d.debug_chains_healed() == 0 and d.debug_ring_recleared() >= 1    # block 30
        assert paths["plan"] == 69, paths
    else:
        assert paths["plan_redone"] >= 1 and 1 <= paths["chain_rerun"] <= 6, paths   # the six batches that were in flight behind batch 30, and no more:
        assert d.debug_chains_healed() == 1                    # ... the chain was finished once and started afresh (it used to stay poisoned: every
                                                               # later batch a refused first attempt plus a synchronous second one, chain_rerun 39)
        assert d.debug_ring_recleared() >= 2                   # blocks 30 and 31 (block 31's re-initialisation went out with batch 33)
        assert paths["plan"] >= 60, paths
    d.close()


def test_async_chain_without_a_sidecar():
    """frame_offsets = NULL + ASYNC: every batch scans its own record boundaries (on the scan stream, into an offsets buffer the
    batch owns) and is chained on the device like the others; an error in the middle still re-runs the batches behind it."""
    w = synth.cfg2()
    buf, offs = w.fill(2 << 20)
    pieces = _cut(buf, offs, 6, seed=17)
    paths = _run_chain(w, pieces, sidecar=False)
    assert paths["redone"] == 0 and paths["chain_rerun"] == 0 and paths["plan"] == 6, paths
    w = synth.cfg3()
    buf, offs = w.fill(2 << 20)
    pieces = _cut(buf, offs, 5, seed=23)
    b2, o2 = pieces[2]
    k = int(o2[len(o2) // 2])
    while b2[k + 30] != ord("I"):
        k = int(o2[np.searchsorted(o2, k, side="right")])
    b2[k + 31:k + 35] = 0xEE     # an Insert for a relation id nobody registered
    paths = _run_chain(w, pieces, sidecar=False)


@pytest.mark.parametrize("asynchronous", [True, False])
def test_decode_enqueued_behind_its_boundary_scan(asynchronous):
    """frame_offsets = NULL on a fixed-width stream: from the second batch on the decode is enqueued BEHIND the batch's boundary scan —
    grids sized by a bound taken from the bytes per frame of the batch before, the frame count read on the device
    (DecParams.nframes_dev, plan.hip). Batches of very different sizes; a batch of one-row transactions right after batches of
    1000-row ones (half again as many frames per byte: beyond the bound, decoded again with the count in hand); a batch whose bytes
    stop being frames in the middle (the scan's malformed-header rule: the rest is one frame, which the decoder rejects — and the
    scan's tiles behind it guessed wrong, which the decode kernels must notice on the device). ASYNC and one finished batch per
    call; every batch against the oracle."""
    from etl_amd.decoder import Decoder
    from oracle import oracle
    w = synth.cfg2()
    w2 = synth.Workload([synth.table_fixed()], 0xE71B0B, rows_per_txn=1, start_lsn=0x9000000, name="one_row_transactions")

    def cut(buf, offs, edges):
        out = []
        for a, b in zip(edges[:-1], edges[1:]):
            o = offs[a:b + 1].astype(np.int64)
            out.append((buf[int(o[0]):int(o[-1])].copy(), (o - o[0]).astype(np.uint32)))
        return out
    buf, offs = w.fill(2 << 20)
    nf = len(offs) - 1
    pieces = cut(buf, offs, [0, nf // 64, nf // 32, nf])
    buf, offs = w2.fill(3 << 20)
    pieces.append((buf.copy(), offs.astype(np.uint32)))           # batch 3: beyond the bound
    buf, offs = w2.fill(3 << 19)
    nf = len(offs) - 1
    pieces += cut(buf, offs, [0, 700, nf // 2, nf])                # batches 4 (tiny), 5 (damaged below), 6
    b5, o5 = pieces[5]
    k = len(o5) // 2
    b5[int(o5[k])] = 0x00
    pieces[5] = (b5, np.concatenate([o5[:k + 1], o5[-1:]]).astype(np.uint32))
    o, d = oracle.Oracle(), Decoder(0)
    w.register(o)
    w.register(d)
    dev = DevBufs(pieces)
    fl = abi.F_INPUT_ON_DEVICE | abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL | (abi.F_ASYNC if asynchronous else 0)
    got = []
    for (p, n, po, nfr) in dev.items:
        got.append(d.decode_device(p, n, None, 0, fl))
    for i, ((pb, po), b) in enumerate(zip(pieces, got)):
        rb = o.decode(pb, po)
        rc = b.sync() if asynchronous else b.rc
        assert (rb.err_code != 0) == (rc != 0), f"batch {i}: oracle error {rb.err_code} vs rc {rc} ({b.error})"
        if rb.err_code:
            assert (b.error.code, b.error.frame_index) == (rb.err_code, rb.err_frame), f"batch {i}"
        diff = rb.host_batch().diff(b.host())
        assert not diff, f"batch {i}: {diff[:6]}"
        b.close()
    chained, redone = d.debug_scan_chained()
    d.close()
    if os.environ.get("ETLG_SCAN_CHAIN", "1") != "0":
        assert chained >= 4 and redone >= 1, (chained, redone)


def test_more_batches_in_flight_than_the_result_ring_holds():
    """The contract asks for fewer than 32 ASYNC batches in flight per context (their result blocks live in a ring of 32). A caller that
    queues 75 before its first sync still gets every batch right: the library finishes the oldest ones itself when the ring is about to
    lap (nothing enforced the limit; a slot's next user could run on a block a batch in flight still wrote)."""
    w = synth.cfg2()
    buf, offs = w.fill(3 << 20)
    pieces = _cut(buf, offs, 75, seed=41)
    paths = _run_chain(w, pieces)
    assert paths["plan"] >= 75 and paths["redone"] == 0, paths
    w = synth.cfg3()
    buf, offs = w.fill(3 << 20)
    pieces = _cut(buf, offs, 70, seed=43)
    b2, o2 = pieces[40]
    k = int(o2[len(o2) // 2])
    while b2[k + 30] != ord("I"):
        k = int(o2[np.searchsorted(o2, k, side="right")])
    b2[k + 31:k + 35] = 0xEE     # an Insert for a relation id nobody registered: an error in the middle, the batches behind it decoded again
    _run_chain(w, pieces)


def test_deferred_scan_is_collected_by_whatever_comes_next():
    """An ASYNC batch without a sidecar returns with its boundary scan in flight; the decode is enqueued by the next call on the
    context — another decode, a control-plane call, a frame-tag query, the batch's own sync, its free, or the context's destroy —
    and the result is the oracle's every time."""
    from etl_amd.decoder import Decoder
    from oracle import oracle
    w = synth.cfg2()
    buf, offs = w.fill(1 << 20)
    pieces = _cut(buf, offs, 4, seed=29)
    dev = DevBufs(pieces)
    for what in ("sync", "control_call", "frame_tags", "free", "destroy", "sync_decode"):
        o, d = oracle.Oracle(), Decoder(0)
        w.register(o)
        w.register(d)
        want = [o.decode(b, f) for b, f in pieces]
        p0, n0, _, nf0 = dev.items[0]
        b0 = d.decode_device(p0, n0, None, 0, FLAGS)            # scan in flight, decode not enqueued
        if what == "sync":
            pass
        elif what == "control_call":
            assert d.table_state(999, abi.TS_READY, 0) == 0     # any call that may change what a batch decodes against
        elif what == "frame_tags":
            t = d.frame_tags(pieces[1][0], pieces[1][1])
            assert len(t) == len(pieces[1][1]) - 1
        elif what == "free":
            b0.close()
            b0 = None
        elif what == "destroy":
            d.close()
            d = None
            b0 = None
        elif what == "sync_decode":                              # a synchronous decode behind it: both are decoded, in order
            p1, n1, po1, nf1 = dev.items[1]
            b1 = d.decode_device(p1, n1, po1, nf1, abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL)
            assert b1.rc == 0 and not want[1].host_batch().diff(b1.host())
            b1.close()
        if b0 is not None:
            assert b0.sync() == 0
            assert b0.view().n_frames == nf0 and not want[0].host_batch().diff(b0.host())
            b0.close()
        if d is not None:
            # the context carries on from the right state
            if what != "sync_decode":
                p1, n1, _, nf1 = dev.items[1]
                b1 = d.decode_device(p1, n1, None, 0, FLAGS)
                assert b1.sync() == 0 and not want[1].host_batch().diff(b1.host())
                b1.close()
            d.close()


@pytest.mark.skipif(os.environ.get("ETLG_SIMT_RUN") == "1", reason="full-size batches: MI355X only")
def test_async_chain_full_size_two_streams():
    """BASELINE-size (64 MiB) cfg2 batches, 40 of them back to back with 12 in flight, every arena byte for byte against the
    oracle. At this size two consecutive kernels really are on the chip together for tens of microseconds (the 70-batch test
    above finishes each kernel before the next starts to matter): what the look-back buffers' rotation, the late carried state
    and the ordering of the result ring's re-initialisation behind the result copies have to get right."""
    from etl_amd.decoder import Decoder
    from oracle import oracle
    w = synth.cfg2()
    pieces = [w.fill(64 << 20) for _ in range(4)]
    dev = DevBufs(pieces)
    o, d = oracle.Oracle(), Decoder(0)
    w.register(o)
    w.register(d)
    inflight, done, n = [], 0, 40
    for k in range(n + 1):
        if k < n:
            p, nb, po, nf = dev.items[k % 4]
            inflight.append(d.decode_device(p, nb, po, nf, FLAGS))
        while done < len(inflight) and (len(inflight) - done > 12 or k == n):
            b = inflight[done]
            rb = o.decode(*pieces[done % 4])
            assert b.sync() == 0 and rb.err_code == 0, (done, b.error, rb.err_desc)
            diff = rb.host_batch().diff(b.host())
            assert not diff, f"batch {done}: {diff[:6]}"
            b.close()
            done += 1
    paths = d.debug_paths()
    assert paths["plan"] == n and paths["chain_rerun"] == 0 and paths["plan_redone"] == 0, paths
    assert d.debug_overlapped() >= n - 6, d.debug_overlapped()
    d.close()


def test_copy_decode_behind_batches_in_flight():
    """etlg_copy_decode with ASYNC stream batches still pending on the context: they are finished first (their carried
    transaction state reaches the context), the copy rows decode inside their own virtual transaction, and the stream goes on
    afterwards from the state the pending batches left — all three against the oracle."""
    from etl_amd.decoder import Decoder
    from oracle import oracle
    w = synth.cfg2()
    buf, offs = w.fill(1 << 20)
    pieces = _cut(buf, offs, 5, seed=41)          # transactions span the cuts
    o, d = oracle.Oracle(), Decoder(0)
    w.register(o)
    w.register(d)
    cols = [("id", 20, False, 1), ("t", 25, True, 0)]
    for t in (o, d):
        t.schema_put(4242, 0, cols)
    so, sd = o.table_ready(4242, 0, [1, 1], [1, 0]), d.table_ready(4242, 0, [1, 1], [1, 0])
    rows = [b"%d\trow %d\n" % (i, i) for i in range(300)]
    rbuf = np.frombuffer(b"".join(rows), dtype=np.uint8)
    roffs = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint32)
    dev = DevBufs(pieces)
    inflight = [d.decode_device(p, n, po, nf, FLAGS) for (p, n, po, nf) in dev.items[:3]]   # nothing synced
    want = [o.decode(*pieces[k]) for k in range(3)]
    gc, rc = d.copy_decode(sd, rbuf, roffs), o.copy_decode(so, rbuf, roffs)
    assert gc.rc == 0 and rc.err_code == 0
    assert not rc.host_batch().diff(gc.host())
    for k, b in enumerate(inflight):
        assert b.sync() == 0
        diff = want[k].host_batch().diff(b.host())
        assert not diff, f"batch {k}: {diff[:6]}"
        b.close()
    for k in (3, 4):                               # the stream continues from the state batch 2 left
        p, n, po, nf = dev.items[k]
        b = d.decode_device(p, n, po, nf, FLAGS)
        rb = o.decode(*pieces[k])
        assert b.sync() == 0 and rb.err_code == 0
        diff = rb.host_batch().diff(b.host())
        assert not diff, f"batch {k}: {diff[:6]}"
        b.close()
    d.close()


FLAGS_DEFAULT = abi.F_INPUT_ON_DEVICE | abi.F_OUTPUT_ON_DEVICE | abi.F_ASYNC     # no NO_CONTROL assertion


def _chain_default_flags(w, pieces, warm, ready=True):
    """`warm` batches one at a time (each synced: the context learns whether the stream carries control frames), the rest
    enqueued back to back and synced afterwards; every batch against the oracle."""
    from etl_amd.decoder import Decoder
    from oracle import oracle
    o, d = oracle.Oracle(), Decoder(0)
    w.register(o, ready=ready)
    w.register(d, ready=ready)
    dev = DevBufs(pieces)
    want = []
    for bb, ff in pieces:
        rb = o.decode(bb, ff)
        want.append((rb, rb.host_batch()))    # (the view lists the schema slots that exist NOW)

    def check(k, b):
        rc = b.sync()
        rb, hb = want[k]
        assert (rb.err_code != 0) == (rc != 0), f"batch {k}: oracle error {rb.err_code} vs rc {rc} ({b.error})"
        if rb.err_code:
            assert (b.error.code, b.error.frame_index) == (rb.err_code, rb.err_frame), f"batch {k}"
        diff = hb.diff(b.host())
        assert not diff, f"batch {k}: {diff[:6]}"
        b.close()

    for k in range(warm):
        p, n, po, nf = dev.items[k]
        check(k, d.decode_device(p, n, po, nf, FLAGS_DEFAULT))
    infl = [d.decode_device(p, n, po, nf, FLAGS_DEFAULT) for (p, n, po, nf) in dev.items[warm:]]
    for k, b in enumerate(infl):
        check(warm + k, b)
    paths = d.debug_paths()
    paths["overlapped"] = d.debug_overlapped()
    paths["ctl_ahead"] = d.debug_ctl_ahead()
    paths["chain_reissued"] = d.debug_chain_reissued()
    d.close()
    return paths


def test_async_is_honoured_without_the_no_control_assertion():
    """Default flags + ASYNC on a stream without Relation / DDL frames: the batches are chained on the device exactly like
    NO_CONTROL ones (optimistic first attempt), none takes the control path."""
    w = synth.cfg2()
    buf, offs = w.fill(3 << 20)
    paths = _chain_default_flags(w, _cut(buf, offs, 8, seed=7), warm=0)
    assert paths["plan"] == 8 and paths["control"] == 0 and paths["chain_rerun"] == 0 and paths["overlapped"] >= 5, paths


def test_async_control_stream_runs_its_pre_pass_ahead():
    """cfg5 (Relation / DDL frames in every batch) with default flags + ASYNC: once the context has seen control frames, the
    control pre-pass of batch k+1 runs ahead on the control stream and its host control plane while batch k is decoded; the
    arenas, the schema slots the Relation messages create and the carried transaction state are the oracle's."""
    w = synth.cfg5()
    buf, offs = w.fill(3 << 20)
    pieces = _cut(buf, offs, 9, seed=13)
    paths = _chain_default_flags(w, pieces, warm=1, ready=False)
    assert paths["ctl_ahead"] == 8 and paths["control"] >= 9 and paths["chain_rerun"] == 0, paths


def test_async_cold_start_on_a_stream_that_begins_with_its_relation_frames():
    """No warm-up: every batch of a cfg5 stream enqueued before the first is synced. The first batch carries the stream's Relation
    frames, so its optimistic attempt ends with the control hint, it takes the control path when it is synced, and the batches behind
    it — enqueued when the context had no schema slot at all — are decoded again from the state it leaves (since round 6: enqueued
    again behind it, chained to its new result, instead of one synchronous second attempt each). Their views list the slots their
    events name (they listed none: found by tools/async_fuzz.py, round 4)."""
    w = synth.cfg5()
    buf, offs = w.fill(1 << 20)
    pieces = _cut(buf, offs, 6, seed=5)
    paths = _chain_default_flags(w, pieces, warm=0, ready=False)
    assert paths["chain_rerun"] + paths["chain_reissued"] >= 1 and paths["control"] >= 1, paths


def test_async_control_stream_with_an_error_in_the_middle():
    """... and with a malformed integer in batch 4: its error is the oracle's, its control-plane effects end at the failing
    frame, and the batches behind it — whose pre-pass and control plane had already run — are decoded again from that state."""
    w = synth.cfg5()
    buf, offs = w.fill(3 << 20)
    pieces = _cut(buf, offs, 9, seed=19)
    b4, o4 = pieces[4]
    k = int(o4[len(o4) // 2])
    while not (b4[k + 30] == ord("I") and b4[k + 31 + 4] == ord("N") and b4[k + 31 + 7] == ord("t")):   # an Insert whose first cell is text-form
        k = int(o4[np.searchsorted(o4, k, side="right")])
    ln = int.from_bytes(b4[k + 39:k + 43].tobytes(), "big")
    assert ln >= 1
    b4[k + 43] = ord("x")       # first byte of the first cell (an integer key in every cfg5 table)
    paths = _chain_default_flags(w, pieces, warm=1, ready=False)
    assert paths["ctl_ahead"] >= 4 and paths["chain_rerun"] >= 1, paths


def test_async_control_stream_with_a_ddl_message_the_host_rejects():
    """A DDL message whose JSON does not parse, in a batch whose pre-pass ran ahead: the host control plane fails on that frame, the
    batch is not run in the chain (it needs the multi-pass kernels with the exact carried state), the batches behind it stop, and
    everything is decoded in order at sync time — error, frame and the events before it as the oracle has them."""
    w = synth.cfg5()
    buf, offs = w.fill(3 << 20)
    pieces = _cut(buf, offs, 7, seed=31)
    hit = False
    for bi in (3, 4, 5):
        bb, oo = pieces[bi]
        for fi in range(len(oo) - 1):
            k = int(oo[fi])
            if bb[k + 30] == ord("M"):
                body = bytes(bb[k:int(oo[fi + 1])])
                j = body.find(b"{")
                assert j > 0
                bb[k + j] = ord("x")
                hit = True
                break
        if hit:
            break
    assert hit
    paths = _chain_default_flags(w, pieces, warm=1, ready=False)
    assert paths["ctl_ahead"] >= 3 and paths["chain_rerun"] >= 1, paths


@pytest.mark.parametrize("hold", ["1", "0"])
def test_control_stream_one_batch_deep_with_a_ddl_message_the_host_rejects(hold):
    """The pipelining depth the Rust batcher runs at (issue k + 1, then collect k) on a stream with control frames, ETLG_CTL_HOLD on and
    off: batch k + 1's control pre-pass goes out before the held batch k is flushed. When k's host control plane rejects a DDL message
    and every batch before k has been collected, k goes to the multi-pass kernels at once — on the decode stream, with the per-frame
    scratch a pre-pass running ahead may be writing (the two were unordered). Error, frame, the events before it and the
    batches behind it must be the oracle's on every rotation of the two scratch sets: the rejected message is tried in batch 3, 4 and 5."""
    import os
    from etl_amd.decoder import Decoder
    from oracle import oracle
    saved = os.environ.get("ETLG_CTL_HOLD")
    os.environ["ETLG_CTL_HOLD"] = hold
    try:
        for target in (3, 4, 5):
            w = synth.cfg5()
            buf, offs = w.fill(3 << 20)
            pieces = _cut(buf, offs, 8, seed=41)
            bb, oo = pieces[target]
            hit = False
            for fi in range(len(oo) - 1):
                k = int(oo[fi])
                if bb[k + 30] == ord("M"):
                    j = bytes(bb[k:int(oo[fi + 1])]).find(b"{")
                    assert j > 0
                    bb[k + j] = ord("x")
                    hit = True
                    break
            if not hit:
                continue
            o, d = oracle.Oracle(), Decoder(0)
            w.register(o, ready=False)
            w.register(d, ready=False)
            dev = DevBufs(pieces)
            want = []
            for pb, pf in pieces:
                rb = o.decode(pb, pf)
                want.append((rb, rb.host_batch()))
                if rb.err_code:
                    break

            def check(k, b):
                rc = b.sync()
                rb, hb = want[k]
                assert (rb.err_code != 0) == (rc != 0), f"target {target} batch {k}: oracle error {rb.err_code} vs rc {rc} ({b.error})"
                if rb.err_code:
                    assert (b.error.code, b.error.frame_index) == (rb.err_code, rb.err_frame), f"target {target} batch {k}"
                diff = hb.diff(b.host())
                assert not diff, f"target {target} batch {k}: {diff[:6]}"
                b.close()

            n = len(want)       # the stream ends at the rejected message (errors are fatal to the apply loop)
            infl = None
            for k in range(n + 1):
                nxt = d.decode_device(*dev.items[k], FLAGS_DEFAULT) if k < len(pieces) else None   # (one batch behind the failing one is issued: its pre-pass is the one that runs ahead)
                if infl is not None:
                    check(k - 1, infl)
                infl = nxt
                if k == n and infl is not None:
                    infl.sync(); infl.close()      # behind the error: whatever it says, it must not hang or crash
            assert want[-1][0].err_code != 0, "the mutated DDL message was meant to fail"
            d.close()
    finally:
        if saved is None:
            os.environ.pop("ETLG_CTL_HOLD", None)
        else:
            os.environ["ETLG_CTL_HOLD"] = saved


@pytest.mark.parametrize("mk,no_ctrl", [(synth.cfg2, True), (synth.cfg3, True), (synth.cfg2, False)])
def test_async_chain_from_pinned_host_buffers(mk, no_ctrl):
    """ETLG_F_ASYNC with HOST input (the Rust batcher's staging ring, crates/etl-gfx950/src/batcher.rs): the bytes and the sidecar
    of every batch are uploaded into a device block of its own on the library's copy stream and the batch joins the device-side
    chain — transactions span the cuts, nothing is synced before everything is enqueued, a ring of three pinned buffers rotates
    only as batches are collected."""
    from etl_amd.decoder import Decoder
    from oracle import oracle
    w = mk()
    buf, offs = w.fill(3 << 20)
    pieces = _cut(buf, offs, 9, seed=5)
    o, d = oracle.Oracle(), Decoder(0)
    w.register(o)
    w.register(d)
    cap = max(len(b) for b, _ in pieces) + 64
    ncap = (max(len(of) for _, of in pieces) + 1) * 4
    ring = [(d.host_alloc(cap), d.host_alloc(ncap)) for _ in range(3)]
    flags = abi.F_OUTPUT_ON_DEVICE | abi.F_ASYNC | (abi.F_NO_CONTROL if no_ctrl else 0)
    inflight, done = [], 0

    def collect():
        nonlocal done
        b = inflight[done]
        rb = o.decode(*pieces[done])
        assert b.sync() == 0 and rb.err_code == 0, (done, b.error)
        diff = rb.host_batch().diff(b.host())
        assert not diff, f"batch {done}: {diff[:6]}"
        b.close()
        done += 1
    for k, (pb, po) in enumerate(pieces):
        if k - done >= len(ring):       # every buffer of the ring is in flight: collect the oldest batch, its buffer comes back
            collect()
        hb, ho = ring[k % len(ring)]
        hb[:len(pb)] = pb
        ho.view(np.uint32)[:len(po)] = po
        inflight.append(d.decode_host_ptr(hb.ctypes.data, len(pb), ho.ctypes.data, len(po) - 1, flags))
    while done < len(inflight):
        collect()
    assert d.debug_staged() == len(pieces)
    paths = d.debug_paths()
    assert paths["chain_rerun"] == 0 and paths["redone"] == 0, paths
    for hb, ho in ring:
        d.host_free(hb); d.host_free(ho)
    d.close()
