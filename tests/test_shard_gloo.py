"""Multi-GPU partitioning logic on CPU: commit-aligned shard planning and the
one collective of the path (all-gather of the 64-byte per-rank header) with
world_size 2 over gloo."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from etl_amd import shard, synth
from tests import pgwire as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_shards_cuts_only_after_commit_and_covers_stream():
    w = synth.cfg3()
    buf, offs = w.fill(2 << 20)
    for n in (1, 2, 4, 8):
        ranges = shard.plan_shards(buf, offs, n)
        assert len(ranges) == n and ranges[0][0] == 0 and ranges[-1][1] == len(offs) - 1
        for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
            assert a1 == b0
        for f0, f1 in ranges[:-1]:
            if f1 > f0:
                assert buf[int(offs[f1 - 1]) + 30] == ord("C")  # ends right after a Commit
        sizes = [int(offs[f1]) - int(offs[f0]) for f0, f1 in ranges]
        assert sum(sizes) == len(buf)
        if n > 1:
            assert max(sizes) < 2 * (len(buf) / n) + (1 << 16)


def test_slice_shard_is_self_contained():
    s = W.Stream()
    for t in range(4):
        s.add(W.begin(0x100 * (t + 1)))
        s.add(W.insert(1, ["%d" % t]))
        s.add(W.commit(0x100 * (t + 1), 0x100 * (t + 1) + 8))
    buf = s.bytes()
    ranges = shard.plan_shards(buf, s.offsets, 2)
    parts = [shard.slice_shard(buf, s.offsets, f0, f1) for f0, f1 in ranges]
    assert b"".join(p[0].tobytes() for p in parts) == buf
    for b, o in parts:
        assert o[0] == 0 and o[-1] == len(b) and b[0] == ord("d")


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # each rank pretends to have decoded its shard: (events, fixed, heap, frames) differ per rank
    hdr = shard.make_header(n_events=1000 + rank, fixed_bytes=24 * (1000 + rank), heap_bytes=4 * rank,
                            n_frames=1000 + rank, payload=(50 * rank, 0, 0))
    g = shard.all_gather_headers(torch.from_numpy(hdr)).numpy()
    lay = shard.global_layout(g)
    np.save(os.path.join(out_dir, f"lay{rank}.npy"),
            np.array([lay["total_events"], lay["total_fixed"], lay["total_heap"], lay["total_frames"],
                      int(lay["event_offsets"][rank]), int(lay["fixed_offsets"][rank]), int(lay["any_error"])]))
    # the grouped, asynchronous variant bench.py uses: 7 batches, 3 headers per collective (last group partial)
    hg = shard.HeaderGatherer(7, 3, "cpu")
    for k in range(7):
        hg.slot(k).copy_(torch.from_numpy(shard.make_header(n_events=10 * k + rank, fixed_bytes=k, heap_bytes=rank, n_frames=10 * k + rank)))
        hg.batch_done(k)
    hg.flush(7)
    hg.wait()
    rows = np.stack([hg.headers_of(k).numpy() for k in range(7)])   # [batch, rank, 8]
    np.save(os.path.join(out_dir, f"grp{rank}.npy"), rows)
    dist.destroy_process_group()


def test_header_all_gather_world2_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    l0, l1 = np.load(tmp_path / "lay0.npy"), np.load(tmp_path / "lay1.npy")
    assert list(l0[:4]) == list(l1[:4]) == [2001, 24 * 2001, 4, 2001]
    assert (l0[4], l0[5]) == (0, 0) and (l1[4], l1[5]) == (1000, 24000)  # rank order == LSN order
    assert l0[6] == 0 and l1[6] == 0
    # grouped asynchronous gather: every rank sees every rank's header of every batch, in rank order
    g0, g1 = np.load(tmp_path / "grp0.npy"), np.load(tmp_path / "grp1.npy")
    assert np.array_equal(g0, g1) and g0.shape == (7, 2, 8)
    for k in range(7):
        for r in range(2):
            assert list(g0[k, r, [1, 2, 3, 7]]) == [10 * k + r, k, r, 10 * k + r]
