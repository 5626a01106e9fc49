"""Sharded decode == single decode. The reference decodes one ordered stream in one task
(crates/etl/src/replication/apply.rs:1210-1336; ordinals :942-963; Relation / DDL handling :2160-2276, 2363-2440); the
multi-GPU path cuts the stream after Commit frames, broadcasts the control frames of earlier shards, decodes every shard on
its own context and reassembles by concatenation (etl_amd/shard.py). Here: one cfg3 stream (TEXT / NUMERIC rows, heap
references to move) and one cfg5 stream (Relation + DDL messages in the middle: later shards decode against schema slots
created by frames they never saw) are cut into 2 / 4 / 8 shards, and the concatenation must equal the one-context decode
byte for byte — every event column incl. tx_ordinal / commit_lsn / schema_slot, RelationEvents, both arenas.

CPU tier: on oracle contexts, in-process and over a world_size-2 gloo group with the real collectives (control all-gather,
header all-gather, arena all-gather). GPU tier (also run on the emulator build by the CPU suite): the same on Decoder
contexts, i.e. through the C ABI and the device kernels."""
import os
import socket
import sys

import numpy as np
import pytest

from etl_amd import shard, synth
from etl_amd.view import HostBatch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STREAMS = {"cfg3": (synth.cfg3, 3 << 20), "cfg5": (synth.cfg5, 6 << 20)}


def _stream(name):
    mk, nbytes = STREAMS[name]
    w = mk()
    buf, offs = w.fill(nbytes)
    return w, buf, offs


def _host(b):
    return b.host_batch() if hasattr(b, "host_batch") else b.host()


def _check(make_ctx, w, buf, offs, n):
    one = make_ctx()
    ref = one.decode(buf, offs)
    assert (getattr(ref, "rc", None) or getattr(ref, "err_code", 0)) == 0
    parts = shard.decode_sharded(make_ctx, buf, offs, n)
    sizes = [int(_host(b).n_frames) for _, b in parts]
    assert sum(sizes) == len(offs) - 1 and (n == 1 or max(sizes) < len(offs) - 1)
    got = HostBatch.concat([_host(b) for _, b in parts])
    diff = _host(ref).diff(got)
    assert not diff, diff[:6]
    return parts


@pytest.mark.parametrize("name", sorted(STREAMS))
@pytest.mark.parametrize("n", [2, 4, 8])
def test_sharded_decode_equals_single_decode_oracle(name, n):
    from oracle import oracle
    w, buf, offs = _stream(name)

    def mk():
        o = oracle.Oracle()
        w.register(o, ready=not w.cfg.emit_relations)
        return o
    parts = _check(mk, w, buf, offs, n)
    if name == "cfg5":   # the stream really carries control frames past the first shard
        tags = shard.frame_tags(buf, offs)
        f0 = shard.plan_shards(buf, offs, n)[1][0]
        assert np.any((tags[f0:] == ord("R")) | (tags[f0:] == ord("M")))
        assert np.any(tags == ord("M")), "the cfg5 sample must include a DDL message"


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(STREAMS))
@pytest.mark.parametrize("n", [2, 8])
def test_sharded_decode_equals_single_decode_device(name, n):
    from etl_amd.decoder import Decoder
    w, buf, offs = _stream(name)
    made = []

    def mk():
        d = Decoder(0)
        w.register(d, ready=not w.cfg.emit_relations)
        made.append(d)
        return d
    _check(mk, w, buf, offs, n)
    tags = made[0].frame_tags(buf, offs)   # the device classification plans the same cuts
    assert shard.plan_shards(None, offs, n, tags=tags) == shard.plan_shards(buf, offs, n)
    # ... and so does the library's own planner (etlg_shard_plan: classification + cut search on the device), for every shard count —
    # incl. more shards than the stream has Commits (empty ranges, cuts that never go back)
    for m in (1, 2, 3, 5, 8, 64, 1000):
        assert made[0].shard_plan(buf, offs, m) == shard.plan_shards(buf, offs, m), m
    few, fo = shard.slice_shard(buf, offs, 0, 40)          # a stream that ends inside a transaction, and one without any Commit
    few, fo = np.ascontiguousarray(few), np.ascontiguousarray(fo)
    for m in (2, 7):
        assert made[0].shard_plan(few, fo, m) == shard.plan_shards(few, fo, m), m
    t = shard.frame_tags(few, fo)
    k = int(np.flatnonzero(t == ord("B"))[0])
    j = k + 1 + int(np.flatnonzero(t[k + 1:] == ord("C"))[0]) if np.any(t[k + 1:] == ord("C")) else len(t)
    nob, noo = shard.slice_shard(few, fo, k, j)
    assert made[0].shard_plan(np.ascontiguousarray(nob), np.ascontiguousarray(noo), 3) == shard.plan_shards(nob, noo, 3) == [(0, 0), (0, 0), (0, j - k)]
    for d in made:
        d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(STREAMS))
def test_control_stream_extracted_on_the_device(name):
    """etlg_control_stream (the device-side extraction the multi-GPU bench uses: nothing of a shard but its control frames leaves
    the device) == shard.control_stream (numpy on the host), for every shard of 1 / 3 / 8-way cuts: same frames, same bytes, and the
    last tag of a commit-aligned range is 'C'. cfg3 has no control frame after its priming: the stream is empty."""
    from etl_amd.decoder import Decoder
    w, buf, offs = _stream(name)
    d = Decoder(0)
    seen = 0
    for n in (1, 3, 8):
        for f0, f1 in shard.plan_shards(buf, offs, n):
            if f1 == f0:
                continue
            b, oo = shard.slice_shard(buf, offs, f0, f1)
            b, oo = np.ascontiguousarray(b), np.ascontiguousarray(oo, dtype=np.uint32)
            want_b, want_o = shard.control_stream(b, oo)
            got_b, got_o, last = d.control_stream(b.ctypes.data, len(b), oo.ctypes.data, len(oo) - 1, on_device=False)
            assert last == ord("C")
            assert np.array_equal(got_o, want_o) and np.array_equal(got_b, want_b), (name, n, f0)
            seen += len(got_o) - 1
    assert (seen > 0) == (name == "cfg5")
    # a range that starts inside a transaction and ends inside another: control frames without their Begin / Commit travel alone
    tags = shard.frame_tags(buf, offs)
    ctrl = np.flatnonzero((tags == ord("R")) | (tags == ord("M")))
    if len(ctrl):
        i = int(ctrl[len(ctrl) // 2])
        b, oo = shard.slice_shard(buf, offs, i, min(i + 3, len(offs) - 1))
        b, oo = np.ascontiguousarray(b), np.ascontiguousarray(oo, dtype=np.uint32)
        want_b, want_o = shard.control_stream(b, oo)
        got_b, got_o, _last = d.control_stream(b.ctypes.data, len(b), oo.ctypes.data, len(oo) - 1, on_device=False)
        assert np.array_equal(got_o, want_o) and np.array_equal(got_b, want_b)
    d.close()


def test_control_stream_is_what_later_shards_need():
    """Without the broadcast a later shard decodes against a stale cache: the test that the test is meaningful."""
    from oracle import oracle
    w, buf, offs = _stream("cfg5")
    ranges = shard.plan_shards(buf, offs, 4)
    f0, f1 = ranges[-1]
    o = oracle.Oracle()
    w.register(o, ready=not w.cfg.emit_relations)
    b, oo = shard.slice_shard(buf, offs, f0, f1)
    r = o.decode(np.ascontiguousarray(b), oo)
    assert r.err_code != 0   # rows of tables whose Relation message lives in an earlier shard


def _worker(rank, world, port, out_dir, name):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, buf, offs = _stream(name)                      # every rank walks the same deterministic stream ...
    f0, f1 = shard.plan_shards(buf, offs, world)[rank]  # ... and owns one contiguous, commit-aligned range of it
    mine = shard.control_stream(buf, offs, f0, f1)
    streams = shard.all_gather_control(mine)           # collective 1: the control frames of every shard
    o = oracle.Oracle()
    w.register(o, ready=not w.cfg.emit_relations)
    shard.replay_control(o, streams[:rank])
    o.reset_stream_state()
    b, oo = shard.slice_shard(buf, offs, f0, f1)
    res = o.decode(np.ascontiguousarray(b), oo)
    assert res.err_code == 0
    hb = res.host_batch()
    hdr = shard.make_header(hb.n_events, len(hb.fixed), len(hb.heap), hb.n_frames, hb.payload_bytes)
    lay = shard.global_layout(shard.all_gather_headers(torch.from_numpy(hdr)).numpy())   # collective 2: the headers
    signed = {np.uint8: np.uint8, np.uint32: np.int32, np.uint64: np.int64}   # torch has no unsigned 32 / 64-bit dtypes on every build
    arrays = {n: torch.from_numpy(np.ascontiguousarray(getattr(hb, n)).view(signed[dt])) for n, dt in shard.ARENA_ARRAYS}
    g, lens = shard.all_gather_arenas(arrays)          # collective 3 (optional in production): the arenas
    lens = lens.numpy()
    # every rank now holds the LSN-ordered result: rebuild it from the gathered arrays and compare with a single decode
    parts = []
    names = [n for n, _ in shard.ARENA_ARRAYS]
    for r in range(world):
        a = {n: g[n][r, :lens[r, k]].numpy().view(dt) for k, (n, dt) in enumerate(shard.ARENA_ARRAYS)}
        parts.append(HostBatch(int(lens[r, 0]), 0, (0, 0, 0), a["kind"], a["flags"], a["table_id"], a["schema_slot"], a["start_lsn"],
                               a["commit_lsn"], a["tx_ordinal"], a["body_off"], a["fixed"], a["heap"], hb.slots))
    # slot descriptors come from the context that has seen every control frame: the last rank's (broadcast in production)
    slots = [None]
    if rank == world - 1:
        slots = [hb.slots]
    dist.broadcast_object_list(slots, src=world - 1)
    for p in parts:
        p.slots = slots[0]
    got = HostBatch.concat(parts)
    one = oracle.Oracle()
    w.register(one, ready=not w.cfg.emit_relations)
    ref = one.decode(buf, offs).host_batch()
    got.n_frames, got.payload_bytes = lay["total_frames"], ref.payload_bytes
    diff = ref.diff(got)
    ok = ((not diff) and lay["total_events"] == ref.n_events and lay["total_fixed"] == len(ref.fixed)
          and lay["total_heap"] == len(ref.heap) and int(lay["event_offsets"][rank]) == sum(int(lens[r, 0]) for r in range(rank)))
    np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.array([int(ok), len(diff)]))
    dist.destroy_process_group()


@pytest.mark.parametrize("name", sorted(STREAMS))
def test_sharded_decode_world2_gloo(tmp_path, name):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path), name), nprocs=2, join=True)
    for r in range(2):
        ok = np.load(tmp_path / f"ok{r}.npy")
        assert ok[0] == 1, f"rank {r}: {ok}"
