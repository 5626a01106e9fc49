"""Multi-GPU partitioning of a WAL stream.

The decode path shards by contiguous, commit-aligned ranges of CopyData frames:
every rank decodes its own range with its own context (transaction state is
shard-local because a cut is only made after a Commit), schemas are replicated
to every rank by the host, and there is no data-path collective. The single
collective is an all-gather of one fixed-size header per rank; because rank
order == LSN order, the exclusive prefix of the gathered counts is each
shard's position in the global, LSN-ordered event sequence (reassembly is
concatenation, no merge).
"""
import numpy as np

# The header is the first 64 bytes of the device result block of a batch
# (etlg_batch_header_to_device): 8 little-endian u64.
HEADER_FIELDS = ("first_err", "n_events", "fixed_bytes", "heap_bytes", "pay_insert", "pay_update", "pay_delete", "n_frames")
NO_ERR = -1  # first_err == ~0 read as int64


def plan_shards(buf, offsets, n_shards):
    """Cut frames [0, nframes) into n_shards contiguous ranges that end right after a
    Commit ('C') frame, balanced by bytes. Returns [(f0, f1)] (f1 exclusive)."""
    offsets = np.asarray(offsets, dtype=np.int64)
    nfr = len(offsets) - 1
    if nfr == 0:
        return [(0, 0)] * n_shards
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    starts = offsets[:-1]
    lens = offsets[1:] - starts
    tag_ok = lens >= 31
    tags = np.zeros(nfr, dtype=np.uint8)
    idx = np.flatnonzero(tag_ok & (a[np.minimum(starts + 5, len(a) - 1)] == ord("w")))
    tags[idx] = a[starts[idx] + 30]
    commit_ends = np.flatnonzero(tags == ord("C")) + 1  # candidate cut points (frame index after a Commit)
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

    def __init__(self, nbatches, group, device, world=None, pg=None):
        import torch
        import torch.distributed as dist
        self.G = max(1, int(group))
        self.pg = pg
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
