"""Multi-GPU partitioning of a WAL stream.

The reference decodes ONE ordered stream in one task (crates/etl/src/replication/apply.rs:1210-1336); what a frame
decodes to depends on two pieces of running state: the open transaction (commit_lsn, next ordinal: apply.rs:942-963) and
the shared table cache that Relation / DDL-message frames update (apply.rs:2160-2276, 2363-2440). The decode path shards by
contiguous ranges of CopyData frames and removes both dependencies:

  * cuts are only made right after a Commit ('C') frame (`plan_shards`), so the transaction state is shard-local;
  * the rare control frames are BROADCAST: every shard extracts the transactions of its range that carry a Relation or a
    DDL message, reduced to {Begin, control frames, Commit} (`control_stream`, a few hundred bytes), the ranks all-gather
    them, and every rank replays those of the ranks before it on its own context before it decodes (`replay_control`) —
    same schema slots in the same order on every rank, so the arenas reference identical slot ids;
  * rank order == LSN order: one all-gather of a 64-byte header per rank gives every rank the exclusive prefix of
    (events, fixed bytes, heap bytes) of its shard in the LSN-ordered result (`global_layout`), and — optionally — one
    padded all-gather per arena array puts the whole result on every rank (`ArenaGatherer`); reassembly is concatenation
    plus that prefix on body offsets and heap references (`etl_amd.view.HostBatch.concat`), never a merge.
"""
import numpy as np

# The header is the first 64 bytes of the device result block of a batch
# (etlg_batch_header_to_device): 8 little-endian u64.
HEADER_FIELDS = ("first_err", "n_events", "fixed_bytes", "heap_bytes", "pay_insert", "pay_update", "pay_delete", "n_frames")
NO_ERR = -1  # first_err == ~0 read as int64


def frame_tags(buf, offsets):
    """pgoutput tag of every frame, on the host (numpy gather). Streams that are already in HBM use the device
    classification instead (Decoder.frame_tags / etlg_frame_tags): same values for well-formed frames."""
    offsets = np.asarray(offsets, dtype=np.int64)
    nfr = len(offsets) - 1
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    tags = np.zeros(nfr, dtype=np.uint8)
    if nfr == 0:
        return tags
    starts = offsets[:-1]
    lens = offsets[1:] - starts
    outer = a[np.minimum(starts + 5, len(a) - 1)]
    idx = np.flatnonzero((lens >= 31) & (outer == ord("w")))
    tags[idx] = a[starts[idx] + 30]
    tags[(lens >= 23) & (outer == ord("k"))] = ord("k")
    return tags


def plan_shards(buf, offsets, n_shards, tags=None, decoder=None):
    """Cut frames [0, nframes) into n_shards contiguous ranges that end right after a
    Commit ('C') frame, balanced by bytes. Returns [(f0, f1)] (f1 exclusive).
    `decoder` (an etl_amd.Decoder): the cut is made by the library — etlg_shard_plan, what a non-Python host calls; the numpy code below is
    its model (the oracle tier of the tests has no library context) and the two are compared in tests/test_shard_decode.py.
    `tags`: the frames' pgoutput tags if the caller already has them (etlg_frame_tags); `buf` may then be None."""
    if decoder is not None and hasattr(decoder, "shard_plan"):
        return decoder.shard_plan(buf, offsets, n_shards)
    offsets = np.asarray(offsets, dtype=np.int64)
    nfr = len(offsets) - 1
    if nfr == 0:
        return [(0, 0)] * n_shards
    if tags is None:
        tags = frame_tags(buf, offsets)
    commit_ends = np.flatnonzero(np.asarray(tags) == ord("C")) + 1  # candidate cut points (frame index after a Commit)
    total = int(offsets[-1])
    cuts = [0]
    for k in range(1, n_shards):
        target = total * k // n_shards
        if len(commit_ends) == 0:
            cuts.append(cuts[-1])
            continue
        j = int(np.searchsorted(offsets[commit_ends], target))
        j = min(j, len(commit_ends) - 1)
        c = int(commit_ends[j])
        cuts.append(max(c, cuts[-1]))
    cuts.append(nfr)
    return [(cuts[i], cuts[i + 1]) for i in range(n_shards)]


def slice_shard(buf, offsets, f0, f1):
    offsets = np.asarray(offsets, dtype=np.int64)
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    b0, b1 = int(offsets[f0]), int(offsets[f1])
    return a[b0:b1], (offsets[f0:f1 + 1] - b0).astype(np.uint32)


def control_stream(buf, offsets, f0=0, f1=None, tags=None):
    """The control frames of frames [f0, f1) as a tiny stream of their own: for every transaction that holds a Relation
    ('R') or logical-decoding Message ('M') frame, its Begin, those frames in order, and its Commit (frames outside any
    transaction are kept as they are: the decoder reports them exactly as it would in place). Decoding it on a fresh
    context has the same effect on the schema store and the shared table cache as decoding the whole range
    (apply.rs:2160-2276, 2363-2440 are the only writers), at none of the cost. Returns (bytes as np.uint8, offsets as
    np.uint32) — possibly empty."""
    offsets = np.asarray(offsets, dtype=np.int64)
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    nfr = len(offsets) - 1
    f1 = nfr if f1 is None else f1
    if tags is None:
        tags = frame_tags(a, offsets)
    t = np.asarray(tags)[f0:f1]
    ctrl = np.flatnonzero((t == ord("R")) | (t == ord("M")))
    if len(ctrl) == 0:
        return np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.uint32)
    begins = np.flatnonzero(t == ord("B"))
    commits = np.flatnonzero(t == ord("C"))
    keep = set(int(i) for i in ctrl)
    for i in ctrl:
        b = np.searchsorted(begins, i, side="right") - 1
        if b < 0:
            continue                                  # before the first Begin of the range: not in a transaction here
        bi = int(begins[b])
        c = np.searchsorted(commits, bi, side="left")
        ci = int(commits[c]) if c < len(commits) else None
        if ci is not None and ci < i:
            continue                                  # that transaction was closed before the control frame
        keep.add(bi)
        if ci is not None:
            keep.add(ci)
    frames = sorted(keep)
    pieces = [a[int(offsets[f0 + i]):int(offsets[f0 + i + 1])] for i in frames]
    lens = np.array([len(p) for p in pieces], dtype=np.int64)
    return np.concatenate(pieces), np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)


def replay_control(decoder, streams):
    """Applies the control streams of the shards BEFORE this one (in rank order) to `decoder` — a Decoder or anything with
    its decode() — and drops the events. Raises if a replay fails: the shard that owns those frames reports the error."""
    if hasattr(decoder, "shard_replay"):    # the library's own entry point (etlg_shard_replay): what a non-Python host calls
        for buf, offs in streams:
            if len(offs) > 1:
                decoder.shard_replay(buf, offs)
        return
    for buf, offs in streams:
        if len(offs) <= 1:
            continue
        b = decoder.decode(buf, offs)
        rc = getattr(b, "rc", None)
        if rc is None:
            rc = getattr(b, "err_code", 0)
        if rc != 0:
            raise RuntimeError(f"control replay failed: {getattr(b, 'error', None) or rc}")
        if hasattr(b, "close"):
            b.close()


def all_gather_control(stream, group=None):
    """All-gathers every rank's control stream (padded uint8 + lengths; they are tiny) and returns them in rank order."""
    import torch
    import torch.distributed as dist
    buf, offs = stream
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    sizes = torch.tensor([len(buf), len(offs)], dtype=torch.int64, device=dev)
    all_sizes = torch.empty((world, 2), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_sizes, sizes.reshape(1, 2), group=group)
    all_sizes = all_sizes.cpu().numpy()
    mb, mo = int(all_sizes[:, 0].max()), int(all_sizes[:, 1].max())
    pb = torch.zeros(max(mb, 1), dtype=torch.uint8, device=dev)
    po = torch.zeros(max(mo, 1), dtype=torch.int64, device=dev)
    if len(buf):
        pb[:len(buf)] = torch.from_numpy(np.ascontiguousarray(buf)).to(dev)
    po[:len(offs)] = torch.from_numpy(np.asarray(offs, dtype=np.int64)).to(dev)
    gb = torch.empty((world, pb.numel()), dtype=torch.uint8, device=dev)
    go = torch.empty((world, po.numel()), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(gb, pb.reshape(1, -1), group=group)
    dist.all_gather_into_tensor(go, po.reshape(1, -1), group=group)
    gb, go = gb.cpu().numpy(), go.cpu().numpy()
    return [(gb[r, :all_sizes[r, 0]].copy(), go[r, :all_sizes[r, 1]].astype(np.uint32)) for r in range(world)]


ARENA_ARRAYS = (("kind", np.uint8), ("flags", np.uint8), ("table_id", np.uint32), ("schema_slot", np.uint32),
                ("start_lsn", np.uint64), ("commit_lsn", np.uint64), ("tx_ordinal", np.uint64), ("body_off", np.uint64),
                ("fixed", np.uint8), ("heap", np.uint8))


def all_gather_arenas(arrays, group=None):
    """The optional data-path collective of north_star: every rank ends up with every shard's arena arrays.
    `arrays`: dict name -> 1-D torch tensor of this rank's shard (device tensors under nccl, CPU tensors under gloo),
    names as in ARENA_ARRAYS. One all-gather of the lengths, then one padded all_gather_into_tensor per array (RCCL moves
    world x max-shard bytes per array over xGMI). Returns (gathered: name -> [world, pad] tensor, lengths [world, n_arrays])."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    names = [n for n, _ in ARENA_ARRAYS]
    dev = arrays[names[0]].device
    lens = torch.tensor([arrays[n].numel() for n in names], dtype=torch.int64, device=dev)
    all_lens = torch.empty((world, len(names)), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_lens, lens.reshape(1, -1), group=group)
    pads = all_lens.max(dim=0).values.cpu().tolist()
    out = {}
    for k, n in enumerate(names):
        t = arrays[n]
        item = t.element_size()
        pad = max(int(pads[k]), 1) * item
        raw = t.contiguous().view(torch.uint8)           # the collective moves bytes: every backend takes uint8
        src = raw if raw.numel() == pad else torch.cat([raw, torch.zeros(pad - raw.numel(), dtype=torch.uint8, device=dev)])
        g = torch.empty((world, pad), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(g, src.reshape(1, -1), group=group)
        out[n] = g.view(t.dtype)
    return out, all_lens


def decode_sharded(make_context, buf, offsets, n_shards, tags=None):
    """The whole multi-GPU recipe in one process (tests, and the reference for what N ranks do): cut, broadcast the control
    streams, decode every shard on its OWN context. `make_context()` returns a primed Decoder (or oracle wrapper).
    Returns the list of per-shard batches, in LSN order."""
    if tags is None:
        tags = frame_tags(buf, offsets)
    first = make_context()
    ranges = plan_shards(buf, offsets, n_shards, tags=tags, decoder=first)   # (a Decoder plans through the C ABI: etlg_shard_plan)
    ctrl = [control_stream(buf, offsets, f0, f1, tags=tags) for f0, f1 in ranges]
    out = []
    for k, (f0, f1) in enumerate(ranges):
        ctx = first if k == 0 else make_context()
        replay_control(ctx, ctrl[:k])
        ctx.reset_stream_state()
        b, o = slice_shard(buf, offsets, f0, f1)
        out.append((ctx, ctx.decode(np.ascontiguousarray(b), o)))
    return out


def make_header(n_events, fixed_bytes, heap_bytes, n_frames, payload=(0, 0, 0), first_err=NO_ERR):
    return np.array([first_err, n_events, fixed_bytes, heap_bytes, payload[0], payload[1], payload[2], n_frames], dtype=np.int64)


def global_layout(headers):
    """headers: [world, 8] int64 in rank order (== LSN order: shards are contiguous
    stream ranges). Returns each shard's exclusive offsets in the reassembled,
    LSN-ordered event sequence / arenas, plus totals."""
    h = np.asarray(headers, dtype=np.int64).reshape(-1, len(HEADER_FIELDS))
    ev_off = np.concatenate([[0], np.cumsum(h[:, 1])])
    fx_off = np.concatenate([[0], np.cumsum(h[:, 2])])
    hp_off = np.concatenate([[0], np.cumsum(h[:, 3])])
    return dict(event_offsets=ev_off[:-1], fixed_offsets=fx_off[:-1], heap_offsets=hp_off[:-1],
                total_events=int(ev_off[-1]), total_fixed=int(fx_off[-1]), total_heap=int(hp_off[-1]),
                total_frames=int(h[:, 7].sum()), any_error=bool(np.any(h[:, 0] != NO_ERR)))


def all_gather_headers(header, group=None):
    """One collective per step: gathers the fixed-size header of every rank
    (RCCL over xGMI when the tensor is on the GPU, gloo on CPU tensors)."""
    import torch
    import torch.distributed as dist
    t = torch.as_tensor(header) if not isinstance(header, torch.Tensor) else header
    world = dist.get_world_size(group)
    out = torch.empty((world, t.numel()), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.reshape(1, -1), group=group)
    return out


class HeaderGatherer:
    """The collective of the multi-GPU path as bench.py runs it: every batch writes its 64-byte header
    into its own slot; the headers of `group` consecutive batches travel in ONE asynchronous all-gather
    (group x 64 bytes per rank) that overlaps the decode of the following batches and is waited for once,
    at the end. (One collective per batch made the host the bottleneck: ~22 us of c10d / RCCL enqueue per
    call against a ~100 us kernel, and nothing consumes the global layout sooner.)"""

    def __init__(self, nbatches, group, device, world=None, pg=None, fence=None):
        import torch
        import torch.distributed as dist
        self.G = max(1, int(group))
        self.pg = pg
        self.fence = fence   # Decoder.fence: ASYNC header copies travel on a private stream; the collective's stream waits for them here
        self.world = dist.get_world_size(pg) if world is None else world
        self.nslots = ((nbatches + self.G - 1) // self.G + 1) * self.G
        self.headers = torch.zeros((self.nslots, 8), dtype=torch.int64, device=device)
        self.gathered = torch.zeros((self.nslots // self.G, self.world, self.G * 8), dtype=torch.int64, device=device)
        self.works = []

    def slot(self, k):
        """The tensor row batch k writes its header into (e.g. etlg_batch_header_to_device(..., slot.data_ptr()))."""
        return self.headers[k]

    def _gather_group(self, g):
        import torch.distributed as dist
        if self.fence is not None:
            self.fence()
        self.works.append(dist.all_gather_into_tensor(self.gathered[g], self.headers[g * self.G:(g + 1) * self.G].reshape(1, -1),
                                                      group=self.pg, async_op=True))

    def batch_done(self, k):
        """Call after batch k's header copy has been enqueued; issues the group's all-gather when it is complete."""
        if (k + 1) % self.G == 0:
            self._gather_group(k // self.G)

    def flush(self, k_end):
        """Gathers the last, partial group (k_end = one past the last batch issued)."""
        if k_end % self.G:
            self._gather_group(k_end // self.G)

    def wait(self):
        while self.works:
            self.works.pop().wait()

    def headers_of(self, k):
        """[world, 8] headers of batch k in rank order (valid after wait())."""
        return self.gathered[k // self.G][:, (k % self.G) * 8:(k % self.G + 1) * 8]
