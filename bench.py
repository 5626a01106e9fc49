#!/usr/bin/env python
"""bench.py — WAL bytes/s decoded on MI355X (BASELINE.json metric).

A "step" is one pass of the decode hot path over `--inner` (10) consecutive 64 MiB batches of synthetic WAL
(BASELINE.json configs[1]: fixed-width 5 x int4 INSERT tuples, 113-byte CopyData frames) that are already resident in HBM
when the timed region starts; the decoded arenas stay in HBM. Batches rotate through a pool larger than the 256 MiB
Infinity Cache so steps do not re-read cached input. The driver's `--steps 20` is therefore 200 decodes (the timed region
of round 1 was 20 decodes, 1.6 ms).

  python bench.py [--gpus N --steps K --warmup W]

N > 1 is launched by the driver under torch.distributed.run (one rank per GPU). Each rank decodes its own contiguous,
commit-aligned shard of the stream (weak scaling: `--inner` batches per rank per step); the collective of the path is an
all-gather of a 64-byte header per rank per batch (RCCL over xGMI) that gives every rank the LSN-ordered global layout.
`--workload cfg4` is BASELINE configs[3]: ONE 8 GiB stream cut into N commit-aligned shards, device-side boundary scan,
control-frame broadcast, header all-gather and (`--gather-arenas`) the padded arena all-gather.

Rank 0 prints ONE JSON line. Beside the headline it carries one object per other BASELINE config, each with its own
algorithmic bytes and kernel times: `cfg3` (mixed I/U/D, TEXT / NUMERIC), `cfg5` (Relation / DDL messages in the stream,
default flags: the control path), `copy` (table-copy rows), `no_sidecar` (record-boundary scan on the device first), `default_flags` (the headline workload
without the caller's no-control assertion: the optimistic control path), `handoff` (columns / RowBinary / protobuf of a decoded
batch), `cfg4` (with --workload cfg4 or --cfg4-leg).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import etl_amd  # noqa: E402,F401  (first thing, before torch initialises HIP: sets the process's HIP runtime defaults — hardware queues)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Algorithmic bytes of one launch: every input byte read once (frames + the 4-byte offsets sidecar per
# frame) and every arena byte written once (42 bytes of event header columns per event + the fixed row arena + the heap).
# cfg2: 113 + 4 + 42 + 24 = 183 bytes per row.
SIDECAR_BYTES_PER_FRAME = 4
HEADER_BYTES_PER_EVENT = 42     # kind 1, flags 1, table 4, slot 4, start 8, commit 8, ordinal 8, body offset 8
WINDOW = int(os.environ.get("ETLG_BENCH_WINDOW", "24"))   # ASYNC batches in flight per context (the library's result ring holds 32)


def cpu_threads(requested):
    """Threads of the all-cores CPU leg: the cores this process may run on (capped at 64), or --cpu-threads."""
    if requested:
        return max(1, requested)
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return min(n, 64)


def cpu_baseline_threads(w, pool, nthreads, seconds):
    """SURVEY.md §8(d) leg (ii): the oracle on `nthreads` host threads. Every batch of the pool is cut into
    `nthreads` contiguous commit-aligned shards (etl_amd/shard.py: the same cut the multi-GPU path uses, so the
    transaction state is shard-local); thread t decodes shard t of every batch with its own oracle context
    (the library call releases the GIL). Rate = bytes of the pool / wall time of the slowest thread."""
    import threading

    import numpy as np

    from etl_amd import shard
    from oracle import oracle
    work = [[] for _ in range(nthreads)]
    total_bytes = total_frames = 0
    for buf, offs in pool:
        for t, (f0, f1) in enumerate(shard.plan_shards(buf, offs, nthreads)):
            if f1 > f0:
                b, o = shard.slice_shard(buf, offs, f0, f1)
                work[t].append((np.ascontiguousarray(b), o))
        total_bytes += len(buf)
        total_frames += len(offs) - 1
    ctxs = []
    for _ in range(nthreads):
        o = oracle.Oracle(mode=oracle.MODE_FULL)
        w.register(o)
        ctxs.append(o)
    done_frames = [0] * nthreads
    failed = []

    def run(t):
        n = 0
        for b, o in work[t]:
            ctxs[t].reset_stream_state()
            _, _, nf, ec = ctxs[t].decode_timed(b, o)
            if ec != 0 or nf != len(o) - 1:
                failed.append((t, ec, nf))
            n += nf
        done_frames[t] = n

    secs = 0.0
    reps = 0
    best = None
    while secs < seconds or reps < 2:
        ths = [threading.Thread(target=run, args=(t,)) for t in range(nthreads)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        dt = time.perf_counter() - t0
        assert not failed and sum(done_frames) == total_frames, (failed[:3], sum(done_frames), total_frames)
        if reps > 0:   # the first pass warms the allocator arenas of the threads
            best = dt if best is None else min(best, dt)
        secs += dt
        reps += 1
    for o in ctxs:
        o.close()
    return {"value": round(total_bytes / best / 1e9, 4), "unit": "GB/s", "cores": nthreads,
            "events_per_s": round(total_frames / best, 1),
            "sample": f"best of {reps - 1} passes over {len(pool)} batches ({total_bytes} bytes) cut into {nthreads} "
                      "commit-aligned shards, one oracle context per thread"}


class Pipeline:
    """ASYNC decodes of device-resident batches on one context, at most WINDOW in flight (the oldest is synced — its
    own completion event, not the stream — checked and freed before the next one is issued)."""

    def __init__(self, dec, items, flags, check=True):
        self.dec, self.items, self.flags, self.check = dec, items, flags, check
        self.inflight = []
        self.bytes = self.frames = self.out_bytes = self.events = 0
        self.k = 0

    def _retire(self):
        b, nbytes, nfr = self.inflight.pop(0)
        rc = b.sync()
        v = b.view()
        if self.check:
            assert rc == 0 and v.n_frames == nfr, (rc, v.n_events, v.n_frames, nfr, b.error)
        self.out_bytes += HEADER_BYTES_PER_EVENT * v.n_events + v.fixed_bytes + v.heap_bytes
        self.events += v.n_events
        self.bytes += nbytes
        self.frames += nfr
        b.close()

    def issue(self, hdr_ptr=None):
        tb, to, nbytes, nfr = self.items[self.k % len(self.items)]
        self.k += 1
        if len(self.inflight) >= WINDOW:
            self._retire()
        b = self.dec.decode_device(tb.data_ptr(), nbytes, to.data_ptr(), nfr, self.flags)
        if hdr_ptr is not None:
            b.header_to_device(hdr_ptr)
        self.inflight.append((b, nbytes, nfr))

    def drain(self):
        while self.inflight:
            self._retire()


def kernel_table(prof):
    return {k: {"launches": n, "avg_us": 1000.0 * ms / n} for k, (n, ms) in prof.items() if n}


def roofline_of(kern, alg_bytes_per_launch, traffic=None, kern_as_run=None):
    """`kern`: every kernel timed ALONE (the chain kept on one stream); `kern_as_run`: the same launches in the mode the timed region runs
    in (two decode streams: consecutive batches overlap, a launch reads longer — what `rocprofv3 --kernel-trace --stats` of this command
    averages over). `achieved` / `frac` are priced on the as-run duration when there is one (the line must follow from the
    rocprof summary, not from the kernel's best case); the alone figure stays beside it (`alone_us`, `frac_alone`)."""
    dom = max(kern, key=lambda k: kern[k]["avg_us"] * kern[k]["launches"])
    alone_us = kern[dom]["avg_us"]
    run_us = max(alone_us, kern_as_run[dom]["avg_us"]) if kern_as_run and dom in kern_as_run else alone_us
    ach = alg_bytes_per_launch / (run_us * 1e-6) / 1e9
    per_batch = {k: v["avg_us"] * v["launches"] / kern[dom]["launches"] for k, v in kern.items()}
    pipe_us = sum(per_batch.values())
    return {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic,
            "alg_bytes_per_launch": int(alg_bytes_per_launch), "kernel_avg_us": round(run_us, 2),
            "alone_us": round(alone_us, 2), "frac_alone": round(alg_bytes_per_launch / (alone_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
            "pipeline_kernels_us": {k: round(v, 2) for k, v in per_batch.items()},
            "pipeline_sum_us": round(pipe_us, 2),
            "pipeline_frac": round(alg_bytes_per_launch / (pipe_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)}


def kernel_sources_sha():
    """Hash of the kernel sources: a traffic file (tools/traffic.sh) names the sources its counters were taken with, and a file whose
    hash is not this one is not quoted (`roofline.traffic` must not come from an earlier build)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "etl_amd", "csrc")
    for n in sorted(os.listdir(d)):
        if n.endswith((".hip", ".h", ".inc", ".cpp")):
            h.update(n.encode()); h.update(open(os.path.join(d, n), "rb").read())
    return h.hexdigest()[:16]


def traffic_of(path, kernel, batch_mib=None):
    """hbm_bytes_per_launch of a traffic file if it was taken for THIS kernel with THESE sources, else None."""
    if not path or not os.path.exists(path):
        return None
    t = json.load(open(path))
    if t.get("kernel") != kernel or t.get("sources_sha") != kernel_sources_sha():
        return None
    if batch_mib is not None and t.get("batch_mib") not in (None, batch_mib):
        return None
    return t.get("hbm_bytes_per_launch")


def roofline_with_traffic(kern, alg, name):
    """roofline of a side leg, with the HBM traffic of profiles/traffic_<name>.json when that file was taken for this kernel and this build."""
    dom = max(kern, key=lambda k: kern[k]["avg_us"] * kern[k]["launches"])
    return roofline_of(kern, alg, traffic_of(os.path.join(ROOT, "profiles", f"traffic_{name}.json"), dom))


def to_device(pool, dev):
    import numpy as np
    import torch
    d = [(torch.from_numpy(b).to(dev), torch.from_numpy(o.view(np.int32)).to(dev), len(b), len(o) - 1) for b, o in pool]
    torch.cuda.synchronize()
    return d


def deferred_cells(dec, item, extra_flags=0):
    """Share of the row cells of one batch that the kernels hand back DEFERRED (the host finishes them: json, arrays, the
    float / temporal texts the device rule does not settle) — counted from the arena's 2-bit cell states, outside any timed
    region ("the deferred fraction is reported in the bench line")."""
    import numpy as np

    from etl_amd import abi
    tb, to, nbytes, nfr = item
    b = dec.decode_device(tb.data_ptr(), nbytes, to.data_ptr(), nfr, abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL | extra_flags)
    hb = b.host()
    b.close()
    cells = deferred = 0
    for si, slot in enumerate(hb.slots):
        n = len(slot.cols)
        for kind, full in ((ord("I"), True), (ord("U"), True)):
            sel = (hb.kind == kind) & (hb.schema_slot == si)
            base = hb.body_off[sel].astype(np.int64)
            if kind == ord("U"):
                ok = hb.flags[sel] & 3
                base = base + np.where(ok == abi.OLD_FULL, slot.row_bytes_full, np.where(ok == abi.OLD_KEY, slot.row_bytes_key, 0))
            for i in range(n):
                st = (hb.fixed[base + i // 4] >> np.uint8(2 * (i % 4))) & 3
                cells += len(st)
                deferred += int((st == abi.CELL_DEFERRED).sum())
    return {"cells": cells, "deferred": deferred, "frac": round(deferred / cells, 8) if cells else 0.0}


def leg_async(mk, dev_id, dev, cap, npool, nbatches, flags, check, traffic_file=None):
    """One BASELINE config as an ASYNC pipeline on a fresh context: wall-clock rate, then the same again under the
    library's HIP-event profiler for the kernel times."""
    import torch

    from etl_amd import abi
    from etl_amd.decoder import Decoder
    w = mk()
    pool = [w.fill(cap) for _ in range(npool)]
    items = to_device(pool, dev)
    dec = Decoder(dev_id)
    w.register(dec)
    p = Pipeline(dec, items, flags, check)
    for _ in range(WINDOW + 2):       # arena pool priming + warm-up
        p.issue()
    p.drain()
    torch.cuda.synchronize()
    p = Pipeline(dec, items, flags, check)
    t0 = time.perf_counter()
    for _ in range(nbatches):
        p.issue()
    p.drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # kernel times: the chain kept on ONE stream (profile mode 2), i.e. every kernel timed alone — with two decode streams the
    # per-launch durations of consecutive batches overlap and would read as a slower kernel. The overlapped figure and the launch
    # interval of the timed run above are reported beside it.
    nprof = min(nbatches, 60)   # (the profiled passes stay short: a HIP event pair per launch)
    dec.profile(2)
    q = Pipeline(dec, items, flags, check)
    for _ in range(nprof):
        q.issue()
    q.drain()
    torch.cuda.synchronize()
    kern = kernel_table(dec.profile_read())
    dec.profile(False)
    for _try in range(3):   # (an event pair that reads minutes for a 100 us kernel — seen once in six bench runs, gpurun_out/r06g — is measured again)
        dec.profile(True)
        q2 = Pipeline(dec, items, flags, check)
        for _ in range(nprof):
            q2.issue()
        q2.drain()
        torch.cuda.synchronize()
        kern2 = kernel_table(dec.profile_read())
        dec.profile(False)
        if all(kern2[k]["avg_us"] < 20 * kern[k]["avg_us"] + 1000 for k in kern2 if k in kern):
            break
    alg = (q.bytes + SIDECAR_BYTES_PER_FRAME * q.frames + q.out_bytes) / nprof
    dom0 = max(kern, key=lambda k: kern[k]["avg_us"] * kern[k]["launches"])
    traffic = traffic_of(traffic_file, dom0)
    out = {"value": round(p.bytes / dt / 1e9, 3), "unit": "GB/s", "events_per_s": round(p.events / dt, 1),
           "hbm_read_frac": round(p.bytes / dt / 1e9 / HBM_PEAK_GBPS, 5),
           "workload": f"{w.name}: {cap >> 20} MiB batches, device-resident in / out, offsets sidecar, " + ("NO_CONTROL | ASYNC" if flags & abi.F_NO_CONTROL else "default control flags + ASYNC (the caller asserts nothing about Relation / DDL frames: optimistic first attempt, chained on the device like NO_CONTROL batches)"),
           "batches": nbatches, "frames_per_batch": int(p.frames / nbatches), "paths": {**dec.debug_paths(), **dec.debug_rows()},
           "roofline": roofline_of(kern, alg, traffic, kern2), "deferred_cells": deferred_cells(dec, items[0])}
    dom = out["roofline"]["kernel"]
    interval_us = 1e6 * dt / nbatches
    out["roofline"]["two_streams"] = {"kernel_avg_us_overlapped": round(kern2[dom]["avg_us"], 2) if dom in kern2 else None,
                                      "launch_interval_us": round(interval_us, 2),
                                      "effective_GBps": round(alg / (interval_us * 1e-6) / 1e9, 1),
                                      "effective_frac": round(alg / (interval_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)}
    dec.close()
    return out, pool, w


def leg_cfg5(dev_id, dev, cap, npool, passes):
    """BASELINE configs[4]: Relation / DDL messages interleaved with rows of 3 tables, DEFAULT flags + ASYNC (the caller asserts
    nothing about control frames). The first batch of a pass is optimistic and falls to the control path; from then on the context
    knows the stream carries control frames and runs every batch's control pre-pass ahead, on the control stream, beside the decode
    of the batch before (host.cpp ctl_begin). One context for all passes (its output arenas stay pooled); before every pass the
    tables are forgotten and registered again, so the stream — whose schemas evolve with its DDL messages — starts from the same
    state each time."""
    import torch

    from etl_amd import abi, synth
    from etl_amd.decoder import Decoder
    w = synth.cfg5()
    pool = [w.fill(cap) for _ in range(npool)]
    items = to_device(pool, dev)
    best = None
    kern = {}
    tot_bytes = sum(len(b) for b, _ in pool)
    tot_frames = sum(len(o) - 1 for _, o in pool)
    out_bytes = events = 0
    paths = {}
    dec = Decoder(dev_id)
    FL = abi.F_OUTPUT_ON_DEVICE | abi.F_ASYNC
    ev_first = None
    for rep in range(passes + 2):
        for t in w.tables:
            dec.table_forget(t["rel_id"])
        dec.reset_stream_state()
        w.register(dec, ready=False)
        if rep == passes + 1:
            dec.profile(2)   # every kernel timed alone (the pre-pass of batch k+1 otherwise runs beside the decode of batch k)
        n0 = {**dec.debug_paths(), **dec.debug_rows()}
        a0 = dec.debug_ctl_ahead()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pl = Pipeline(dec, items, FL, True)
        for _ in range(npool):
            pl.issue()
        pl.drain()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert ev_first in (None, pl.events), (ev_first, pl.events)    # every pass decodes the same stream from the same state
        ev_first = pl.events
        if rep == passes + 1:
            kern = kernel_table(dec.profile_read())
        elif rep > 0:
            best = dt if best is None else min(best, dt)
            n1 = {**dec.debug_paths(), **dec.debug_rows()}
            paths = {k: n1[k] - n0[k] for k in n1}
            paths["pre_pass_ahead"] = dec.debug_ctl_ahead() - a0
        out_bytes, events = pl.out_bytes, pl.events
    dec.close()
    alg = (tot_bytes + SIDECAR_BYTES_PER_FRAME * tot_frames + out_bytes) / npool
    # kernel table per BATCH: the control path launches classify / scan / ctrl_list only for batches with control frames
    return {"value": round(tot_bytes / best / 1e9, 3), "unit": "GB/s", "events_per_s": round(events / best, 1),
            "hbm_read_frac": round(tot_bytes / best / 1e9 / HBM_PEAK_GBPS, 5),
            "workload": f"{w.name}: {npool} consecutive {cap >> 20} MiB batches of one stream, device-resident in / out, offsets sidecar, "
                        "default flags + ASYNC (Relation / DDL frames handled by the host control plane; control pre-pass of batch k+1 beside the decode of batch k)",
            "batches": npool, "paths": paths, "roofline": roofline_with_traffic(kern, alg, "cfg5")}


def leg_cfg2_mixed(dev_id, dev, cap, npool, nbatches):
    """The fixed-width plan off its insert-only diet: the cfg2 stream with ONE Update in (a) every batch, (b) every 10th batch
    — an Insert frame's tag byte rewritten to 'U': an Update without an old image has the Insert's layout, and since round 6 the plan
    decodes it itself (round 5: the batch was decoded again by the generic kernel and the ASYNC chain behind it started over) — and
    (c) one Delete by key in every 10th batch (the plan's own since round 6's last session: the key-layout row, priced by the pre-pass
    by its length), (d) one Update WITH a key image in every 10th batch, a shape the plan still gives up on: the give-up path — a
    second attempt on the generic kernel; the batches in flight behind it stand when that attempt leaves the carried transaction state
    the plan had published (`chain_rerun` counts the ones that do not). ASYNC chain of `nbatches` 64 MiB batches, NO_CONTROL,
    device-resident in / out."""
    import numpy as np
    import torch

    from etl_amd import abi, synth
    from etl_amd.decoder import Decoder
    w = synth.cfg2()
    pool = [w.fill(cap) for _ in range(npool)]
    out = {}
    FL = abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL | abi.F_ASYNC
    def with_delete(buf, offs, f):
        """frame f (an Insert) replaced by a Delete of the same row by its key: 'D' rel 'K' 1 column"""
        fr = bytes(buf[offs[f]:offs[f + 1]])
        cell_len = int.from_bytes(fr[39:43], "big")            # the first cell: 't' len bytes at offset 38
        body = b"D" + fr[31:35] + b"K" + (1).to_bytes(2, "big") + fr[38:43 + cell_len]
        nf = b"d" + (4 + 25 + len(body)).to_bytes(4, "big") + fr[5:30] + body
        nb = np.concatenate([buf[:offs[f]], np.frombuffer(nf, dtype=np.uint8), buf[offs[f + 1]:]])
        no = offs.astype(np.int64).copy()
        no[f + 1:] += len(nf) - len(fr)
        return nb, no.astype(np.uint32)

    def with_update_key(buf, offs, f):
        """frame f (an Insert) replaced by an Update of the same row that carries its key image: 'U' rel 'K' 1 column 'N' row — a shape the plan gives up on"""
        fr = bytes(buf[offs[f]:offs[f + 1]])
        cell_len = int.from_bytes(fr[39:43], "big")
        body = b"U" + fr[31:35] + b"K" + (1).to_bytes(2, "big") + fr[38:43 + cell_len] + b"N" + fr[36:]
        nf = b"d" + (4 + 25 + len(body)).to_bytes(4, "big") + fr[5:30] + body
        nb = np.concatenate([buf[:offs[f]], np.frombuffer(nf, dtype=np.uint8), buf[offs[f + 1]:]])
        no = offs.astype(np.int64).copy()
        no[f + 1:] += len(nf) - len(fr)
        return nb, no.astype(np.uint32)

    for name, every, how in (("update_in_every_batch", 1, "U"), ("update_in_every_10th_batch", 10, "U"), ("delete_in_every_10th_batch", 10, "D"),
                             ("update_with_key_in_every_10th_batch", 10, "UK")):
        mixed = []
        for k, (buf, offs) in enumerate(pool):
            b2, o2 = buf.copy(), offs
            if k % every == 0:
                tags = b2[offs[:-1] + 30]
                f = int(np.nonzero(tags == ord("I"))[0][len(offs) // 2])   # an Insert in the middle of the batch
                if how == "U":
                    b2[offs[f] + 30] = ord("U")
                elif how == "D":
                    b2, o2 = with_delete(b2, offs, f)
                else:
                    b2, o2 = with_update_key(b2, offs, f)
            mixed.append((b2, o2))
        items = to_device(mixed, dev)
        dec = Decoder(dev_id)
        synth.cfg2().register(dec)
        pl = Pipeline(dec, items, FL, True)
        for _ in range(WINDOW + 2):
            pl.issue()
        pl.drain()
        torch.cuda.synchronize()
        n0 = {**dec.debug_paths(), **dec.debug_rows(), "chain_spared": dec.debug_chains_spared()}
        pl = Pipeline(dec, items, FL, True)
        t0 = time.perf_counter()
        for _ in range(nbatches):
            pl.issue()
        pl.drain()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n1 = {**dec.debug_paths(), **dec.debug_rows()}
        n1["chain_reissued"] = dec.debug_chain_reissued(); n0.setdefault("chain_reissued", 0)
        n1["chain_spared"] = dec.debug_chains_spared()
        dec.close()
        out[name] = {"value": round(pl.bytes / dt / 1e9, 3), "unit": "GB/s", "us_per_batch": round(1e6 * dt / nbatches, 1), "batches": nbatches,
                     "paths": {k: n1[k] - n0[k] for k in n1 if n1[k] - n0[k]}}
        del items
    out["workload"] = f"cfg2 stream, {cap >> 20} MiB batches, one Insert of the named batches rewritten as an Update without an old image / replaced by a Delete by key / by an Update with its key image; pool of {npool} batches, ASYNC | NO_CONTROL, chain of {nbatches}"
    return out


def leg_cfg1(dev_id, dev, cap, cpu_seconds):
    """BASELINE configs[0], the reference's own CPU-runnable case (crates/etl-benchmarks/src/table_streaming.rs:140-267 drives a
    MemoryDestination with one apply worker): the 1 M-row INSERT-only stream of synth.cfg1 — 1 000 transactions of 1 000 rows of the
    five-int4 table, with its Relation frames — decoded once through the device path (default flags + ASYNC, 64 MiB batches,
    device-resident in / out) and once by the CPU port of the reference's decoder on ONE thread (the single apply worker). Events/s is
    the figure the config is about; the whole stream is the sample."""
    import torch

    from etl_amd import abi, synth
    from etl_amd.decoder import Decoder
    from oracle import oracle
    import numpy as np
    w = synth.cfg1()
    buf, offs = w.fill(160 << 20, max_txns=1000)     # the whole stream (113 MB), then cut behind Commits into batches of <= cap
    tags = np.where(buf[offs[:-1] + 5] == ord("w"), buf[offs[:-1] + 30], 0)
    rows = int((tags == ord("I")).sum())
    assert rows == 1_000_000 and int((tags == ord("B")).sum()) == 1000, rows
    commits = np.nonzero(tags == ord("C"))[0]
    pieces, f0 = [], 0
    while f0 < len(offs) - 1:
        ends = commits[(commits >= f0) & (offs[commits + 1] - offs[f0] <= cap)]
        f1 = int(ends[-1]) + 1
        pieces.append((np.ascontiguousarray(buf[offs[f0]:offs[f1]]), (offs[f0:f1 + 1] - offs[f0]).astype(np.uint32)))
        f0 = f1
    items = to_device(pieces, dev)
    tot_bytes = sum(len(b) for b, _ in pieces)
    tot_frames = sum(len(o) - 1 for _, o in pieces)
    dec = Decoder(dev_id)
    FL = abi.F_OUTPUT_ON_DEVICE | abi.F_ASYNC
    best, events, paths = None, 0, {}
    # One decode of the stream is half a millisecond — a latency, not a rate: a timed pass is LOOPS decodes of the whole
    # stream back to back on one context (each starts with the stream's Relation frames again, as after a reconnect: the control path
    # every time; a pass ends behind a Commit), sized so that it lasts >= 100 ms.
    loops = 1
    for rep in range(5):
        for t in w.tables:
            dec.table_forget(t["rel_id"])
        dec.reset_stream_state()
        w.register(dec, ready=False)
        n0 = {**dec.debug_paths(), **dec.debug_rows()}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pl = Pipeline(dec, items, FL, True)
        for _ in range(len(items) * loops):
            pl.issue()
        pl.drain()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rep == 0:
            loops = max(1, int(0.12 / max(dt, 1e-5)) + 1)   # (the first pass also warms the pools)
            continue
        per = dt / loops
        best = per if best is None else min(best, per)
        n1 = {**dec.debug_paths(), **dec.debug_rows()}
        paths = {k: n1[k] - n0[k] for k in n1}
        paths["stream_decodes_per_timed_pass"] = loops
        events = pl.events // loops
    dec.close()
    secs, passes, cpu_events, cpu_best = 0.0, 0, 0, None
    while secs < cpu_seconds or passes == 0:
        o = oracle.Oracle(mode=oracle.MODE_FULL)    # a fresh context per pass: the stream registers its own tables (Relation frames)
        w.register(o, ready=False)
        ps, cpu_events = 0.0, 0
        for buf, offs in pieces:
            s_, ne, nf, ec = o.decode_timed(buf, offs)
            assert ec == 0 and nf == len(offs) - 1
            ps += s_
            cpu_events += ne
        o.close()
        secs += ps
        passes += 1
        cpu_best = ps if cpu_best is None else min(cpu_best, ps)
    assert cpu_events == events, (cpu_events, events)
    return {"workload": f"{w.name}: 1 000 000 INSERT rows in 1 000 transactions with their Relation frames ({tot_bytes} bytes, {tot_frames} frames, {len(pieces)} batches of <= {cap >> 20} MiB)",
            "rows": rows, "events": events,
            "gpu": {"events_per_s": round(events / best, 1), "value": round(tot_bytes / best / 1e9, 3), "unit": "GB/s", "seconds": round(best, 6), "paths": paths,
                    "how": "default flags + ASYNC, device-resident in / out; a timed pass decodes the whole stream `stream_decodes_per_timed_pass` times back to back (>= 100 ms), best of 4 passes, per decode of the stream"},
            "cpu_baseline": {"events_per_s": round(events / cpu_best, 1), "value": round(tot_bytes / cpu_best / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
                             "seconds": round(cpu_best, 4), "sample": f"the whole stream, best of {passes} passes, single thread (the reference's one apply worker), oracle FULL mode"}}


def leg_wide70(dev_id, dev, nrows, reps):
    """The reference's type-matrix table (crates/etl/tests/replication_stream.rs:184-268): 68 replicated columns, wider than k_cells'
    column masks — which kernel takes it and at what rate. Inserts with every fifth row an Update by key and every eleventh a Delete."""
    import torch

    from etl_amd import abi, synth
    from etl_amd.decoder import Decoder
    buf, offs = synth.type_matrix_stream(nrows, mix=True)
    items = to_device([(buf, offs)], dev)
    dec = Decoder(dev_id)
    synth.type_matrix_register(dec)
    FL = abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL | abi.F_ASYNC
    pl = Pipeline(dec, items, FL, True)
    for _ in range(WINDOW + 2):
        pl.issue()
    pl.drain()
    torch.cuda.synchronize()
    pl = Pipeline(dec, items, FL, True)
    t0 = time.perf_counter()
    for _ in range(reps):
        pl.issue()
    pl.drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dec.profile(2)
    q = Pipeline(dec, items, FL, True)
    for _ in range(8):
        q.issue()
    q.drain()
    torch.cuda.synchronize()
    kern = kernel_table(dec.profile_read())
    dec.profile(False)
    paths = {**dec.debug_paths(), **dec.debug_rows()}
    # the same chain with ETLG_F_FINISH_CELLS: every batch's array cells typed and float cells settled on the device when it is synced
    # (etlg_batch_finish_cells: two more kernels over the arena and two stops per batch) — what a consumer of Cell::Array pays instead of
    # re-parsing 31 of the 68 cells of every row on the host
    dfrac = deferred_cells(dec, items[0])
    dfrac_fin = deferred_cells(dec, items[0], abi.F_FINISH_CELLS)
    pf = Pipeline(dec, items, FL | abi.F_FINISH_CELLS, True)
    for _ in range(4):
        pf.issue()
    pf.drain()
    torch.cuda.synchronize()
    pf = Pipeline(dec, items, FL | abi.F_FINISH_CELLS, True)
    t0 = time.perf_counter()
    for _ in range(max(8, reps // 2)):
        pf.issue()
    pf.drain()
    torch.cuda.synchronize()
    dtf = time.perf_counter() - t0
    dec.close()
    alg = (q.bytes + SIDECAR_BYTES_PER_FRAME * q.frames + q.out_bytes) / 8
    return {"value": round(pl.bytes / dt / 1e9, 3), "unit": "GB/s", "events_per_s": round(pl.events / dt, 1),
            "deferred_cells": dfrac,
            "finish_cells": {"value": round(pf.bytes / dtf / 1e9, 3), "unit": "GB/s", "deferred_cells": dfrac_fin,
                             "note": "ETLG_F_FINISH_CELLS: arrays typed (etlg_array_hdr entries) and floats settled in the arena on the device; what is left DEFERRED is json / jsonb and json arrays"},
            "workload": f"type-matrix table, {len(synth.TYPE_MATRIX)} columns (every scalar class, 31 array columns, json): one batch of {len(buf)} bytes / {len(offs) - 1} frames "
                        f"(avg {len(buf) // (len(offs) - 1)} B), I / U(key) / D(key), NO_CONTROL | ASYNC, device-resident in / out",
            "batches": reps, "paths": paths, "roofline": roofline_with_traffic(kern, alg, "wide70")}


def leg_copy(dev_id, dev, nrows, reps):
    """table-copy rows (COPY text format) -> the Insert arena (rows -> arena in one kernel, k_copy_cells; `paths` says how
    many of the timed batches took it and how many fell back to the row -> frame rewrite). `value` is quoted on escape-heavy rows, the
    same table with text that needs no escapes is reported beside it (`ordinary_text`)."""
    import numpy as np
    import torch

    from etl_amd import synth
    from etl_amd.decoder import Decoder
    return _leg_copy_rows(dev_id, dev, synth.copy_rows(20000, 1), nrows, reps, "escape-heavy text (every backslash escape of the format, ~23 % of the text characters)",
                          clean=_leg_copy_rows(dev_id, dev, synth.copy_rows(20000, 1, clean=True), nrows, reps, "ordinary text (no character that COPY escapes)"))


def _leg_copy_rows(dev_id, dev, base, nrows, reps, what, clean=None):
    import numpy as np
    import torch

    from etl_amd import synth
    from etl_amd.decoder import Decoder
    rows = base * max(1, nrows // len(base))
    buf = np.frombuffer(b"".join(rows), dtype=np.uint8)
    offs = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint32)
    d = Decoder(dev_id)
    d.schema_put(42, 0, synth.COPY_COLS)
    slot = d.table_ready(42, 0, [1] * 10, [1] + [0] * 9)
    tb = torch.from_numpy(buf.copy()).to(dev)
    to = torch.from_numpy(offs.view(np.int32).copy()).to(dev)
    torch.cuda.synchronize()
    for _ in range(2):
        d.copy_decode_device(slot, tb.data_ptr(), tb.numel(), to.data_ptr(), len(rows)).close()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        b = d.copy_decode_device(slot, tb.data_ptr(), tb.numel(), to.data_ptr(), len(rows))
        assert b.rc == 0
        b.close()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the ASYNC form (etlg_copy_decode with ETLG_F_ASYNC): three batches in flight, the oldest synced before the next is issued —
    # the host's per-call work (side inputs, outputs, launch) hides behind the kernels of the batches before
    asy = None
    try:
        from etl_amd import abi
        fl = abi.F_OUTPUT_ON_DEVICE | abi.F_ASYNC
        win = []
        for k in range(3 + 3 * reps):
            if k == 3:
                torch.cuda.synchronize()
                ta = time.perf_counter()
            if len(win) == 3:
                b = win.pop(0)
                assert b.sync() == 0
                b.close()
            win.append(d.copy_decode_device(slot, tb.data_ptr(), tb.numel(), to.data_ptr(), len(rows), flags=fl))
        for b in win:
            assert b.sync() == 0
            b.close()
        torch.cuda.synchronize()
        dta = time.perf_counter() - ta
        asy = {"value": round(3 * reps * len(buf) / dta / 1e9, 3), "unit": "GB/s", "in_flight": 3, "batches": 3 * reps}
    except Exception as e:   # (a leg beside the headline must not take the line down)
        asy = {"error": repr(e)[:200]}
    d.profile(True)
    ob = 0
    for _ in range(reps):
        b = d.copy_decode_device(slot, tb.data_ptr(), tb.numel(), to.data_ptr(), len(rows))
        v = b.view()
        ob += HEADER_BYTES_PER_EVENT * v.n_events + v.fixed_bytes + v.heap_bytes
        b.close()
    torch.cuda.synchronize()
    kern = kernel_table(d.profile_read())
    d.profile(False)
    paths = d.debug_copy()
    d.close()
    # algorithmic bytes: the rows + their offsets read once, the arena written once
    alg = len(buf) + 4 * len(rows) + ob / reps
    out = {"value": round(reps * len(buf) / dt / 1e9, 3), "unit": "GB/s", "rows_per_s": round(reps * len(rows) / dt, 1),
           "workload": f"{len(rows)} COPY text rows of a 10-column mixed table ({len(buf)} bytes), {what}, device-resident, synchronous",
           "paths": paths, "roofline": roofline_with_traffic(kern, alg, "copy_clean" if clean is None and "ordinary" in what else "copy"), "async": asy}
    if clean is not None:
        out["ordinary_text"] = clean   # the same table with text that needs no escapes: what COPY output mostly looks like
    return out


def leg_handoff(dev_id, dev, cap, reps):
    """a decoded arena (device-resident) -> Arrow-layout column buffers (etlg_batch_columns), ClickHouse RowBinary rows
    (etlg_batch_rowbinary) and BigQuery protobuf rows (etlg_batch_protobuf), all left in HBM — for a cfg2 batch (5 x int4) and, under
    "cfg3", for BASELINE's var-len schema (TEXT, NUMERIC as its Display string, timestamptz, uuid; inserts + updates + deletes).
    Rates are quoted in WAL input bytes per second so that they compare with `value`; the host-side hand-off of the same cfg2 arena
    (etl_amd/arrow.py, numpy) is timed beside."""
    import numpy as np
    import torch

    from etl_amd import abi, synth
    from etl_amd.decoder import Decoder

    def one(mk, host_leg):
        w = mk()
        d = Decoder(dev_id)
        w.register(d)
        buf, offs = w.fill(cap)
        tb = torch.from_numpy(buf.copy()).to(dev)
        to = torch.from_numpy(offs.astype(np.uint32).view(np.int32).copy()).to(dev)
        b = d.decode_device(tb.data_ptr(), tb.numel(), to.data_ptr(), len(offs) - 1, abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL)
        assert b.rc == 0
        c0 = b.columns(0, on_device=True)
        nc = int(c0.view.n_cols)
        nullable = [1 if c0.column(i).nullable else 0 for i in range(nc)]   # Nullable() destination columns where the source column is
        c0.close()
        out = {"workload": f"one {cap >> 20} MiB {w.name} batch ({len(offs) - 1} frames), arena device-resident, outputs left in HBM"}
        for name, fn in (("arrow_columns", lambda: b.columns(0, kinds=("I", "U"), on_device=True)),
                         ("rowbinary", lambda: b.rowbinary(0, nullable + [0, 0], abi.CH_REPLACING_MERGE_TREE, on_device=True)),
                         ("protobuf", lambda: b.protobuf(0, on_device=True))):
            r = fn()
            assert name == "arrow_columns" or r.status == abi.RB_OK, "the slot must be encoded on the device"
            r.close()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                r = fn()
                nrows = r.n_rows
                nbytes = int(r.view.n_bytes) if name != "arrow_columns" else sum(int(r.column(i).values_bytes) for i in range(nc))
                r.close()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            out[name] = {"value": round(len(buf) / dt / 1e9, 3), "unit": "GB/s", "ms": round(dt * 1e3, 3), "rows": nrows,
                         "rows_per_s": round(nrows / dt, 1), "out_bytes": nbytes}
        if host_leg:   # the host path the device one replaces: download the arena, numpy gathers per column
            from etl_amd.arrow import rows_to_record_batch
            t0 = time.perf_counter()
            hb = b.host()
            t1 = time.perf_counter()
            rb = rows_to_record_batch(hb, 0)
            t2 = time.perf_counter()
            out["host_numpy"] = {"value": round(len(buf) / (t2 - t0) / 1e9, 3), "unit": "GB/s", "download_ms": round((t1 - t0) * 1e3, 2),
                                 "gather_ms": round((t2 - t1) * 1e3, 2), "rows": rb.num_rows}
        b.close()
        d.close()
        return out

    out = one(synth.cfg2, True)
    out["cfg3"] = one(synth.cfg3, False)
    return out


def leg_pcie(dev_id, dev, cap, nbatches):
    """the end-to-end rate when the boundary hands over HOST buffers — never `value`, reported beside it.
    `in`: pinned host input -> the library's copy stream -> decode, arena left in HBM (ETLG_F_ASYNC from host buffers: a ring of
    three pinned buffers, the upload of batch k+1 beside the decode of batch k). `host_to_host`: the same plus the arena copied back
    into pinned host memory per batch (etlg_batch_download), i.e. what a host-side sink consumes. cfg2 batches, offsets sidecar."""
    import numpy as np
    import torch

    from etl_amd import abi, synth
    from etl_amd.decoder import Decoder
    w = synth.cfg2()
    pool = [w.fill(cap) for _ in range(3)]
    d = Decoder(dev_id)
    w.register(d)
    ring = []
    for b, o in pool:   # each pool batch has a pinned buffer of its own: the ring of the batcher, already filled
        hb, ho = d.host_alloc(len(b) + 64), d.host_alloc(len(o) * 4 + 64)
        hb[:len(b)] = b
        ho.view(np.uint32)[:len(o)] = o
        ring.append((hb, ho, len(b), len(o) - 1))
    flags = abi.F_OUTPUT_ON_DEVICE | abi.F_ASYNC | abi.F_NO_CONTROL
    out = {"workload": f"{w.name}: {cap >> 20} MiB batches from a ring of {len(ring)} pinned host buffers, offsets sidecar, NO_CONTROL | ASYNC"}

    def run(n, download):
        inflight, nbytes, out_bytes = [], 0, 0

        def retire():
            nonlocal out_bytes
            b, nb = inflight.pop(0)
            assert b.sync() == 0, b.error
            v = b.view()
            out_bytes += HEADER_BYTES_PER_EVENT * v.n_events + v.fixed_bytes + v.heap_bytes
            if download:
                assert d.L.etlg_batch_download(d.h, b.h) == 0
            b.close()
        for k in range(n):
            if len(inflight) >= len(ring) - 1:   # a buffer is reused only after its batch has been collected
                retire()
            hb, ho, nb, nf = ring[k % len(ring)]
            inflight.append((d.decode_host_ptr(hb.ctypes.data, nb, ho.ctypes.data, nf, flags), nb))
            nbytes += nb
        while inflight:
            retire()
        torch.cuda.synchronize()
        return nbytes, out_bytes
    for name, download in (("in", False), ("host_to_host", True)):
        run(4, download)
        t0 = time.perf_counter()
        nbytes, out_bytes = run(nbatches, download)
        dt = time.perf_counter() - t0
        out[name] = {"value": round(nbytes / dt / 1e9, 3), "unit": "GB/s", "batches": nbatches, "ms_per_batch": round(1e3 * dt / nbatches, 3),
                     "arena_bytes_per_batch": int(out_bytes / nbatches)}
    out["staged_on_copy_stream"] = d.debug_staged()
    for hb, ho, _, _ in ring:
        d.host_free(hb); d.host_free(ho)
    d.close()
    return out


def leg_no_sidecar(dev_id, items, steps, check):
    """The same cfg2 batches with frame_offsets = NULL: the record-boundary scan runs on the device first (scan.hip).
    Reported beside `value`, never as `value`: the reference's host learns every frame length from its socket codec.
    `value`: ASYNC — the scan of batch k+1 runs on its own stream while batch k is decoded, the host only waits for the frame
    count; `sync_value`: one batch at a time, every call returns a finished batch."""
    import torch

    from etl_amd import abi, synth
    from etl_amd.decoder import Decoder
    # a context of its own, like the other legs (the headline's context keeps 24 output sets of its chain in its pools; on it this leg
    # read 700-810 GB/s where a fresh context reads 1 200: gpurun_out/r06zc, r06v against tools/nosidecar_probe.py)
    dec = Decoder(dev_id)
    synth.cfg2().register(dec)
    out = {}
    for mode, fl in (("sync", abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL), ("async", abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL | abi.F_ASYNC)):
        depth = 8 if mode == "async" else 1
        warm = [dec.decode_device(items[k % len(items)][0].data_ptr(), items[k % len(items)][2], None, 0, fl) for k in range(depth + 1)]
        for wb in warm:   # a full window once, untimed: every batch in flight owns an offsets buffer (grow-only pool)
            if mode == "async":
                wb.sync()
            wb.close()
        def chain():
            nb = 0
            window = []
            for k in range(steps):
                tb, to, nbytes, nfr = items[k % len(items)]
                b = dec.decode_device(tb.data_ptr(), nbytes, None, 0, fl)
                window.append((b, nfr))
                nb += nbytes
                if len(window) >= depth:
                    ob, onf = window.pop(0)
                    rc = ob.sync() if mode == "async" else ob.rc
                    assert not check or (rc == 0 and ob.view().n_frames == onf)
                    ob.close()
            for ob, onf in window:
                rc = ob.sync() if mode == "async" else ob.rc
                assert not check or (rc == 0 and ob.view().n_frames == onf)
                ob.close()
            return nb
        # the rate is taken WITHOUT the library's per-kernel events (they put an event pair around every launch, and on this leg — three
        # streams, a host wait per batch — that costs a third of the rate: 605 against 867 GB/s, gpurun_out/r06t); the kernel times come
        # from a second, untimed chain with the events on
        best = None
        for _rep in range(3):   # (a chain of `steps` batches is a few milliseconds: the best of three)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nb = chain()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            if best is None or t1 - t0 < best[1] - best[0]:
                best = (t0, t1)
        t0, t1 = best
        dec.profile(True)
        chain()
        torch.cuda.synchronize()
        kern = kernel_table(dec.profile_read())
        dec.profile(False)
        out[mode] = (nb / (t1 - t0) / 1e9, kern)
    kern = out["async"][1]
    kb = kern.get("k_bounds", {"launches": 0, "avg_us": 0.0})
    chained = dec.debug_scan_chained()
    dec.close()
    return {"value": round(out["async"][0], 3), "unit": "GB/s", "sync_value": round(out["sync"][0], 3),
            "k_bounds_avg_us": round(kb["avg_us"], 2), "k_bounds_launches_per_batch": round(kb["launches"] / steps, 2),
            "kernels_us": {k: round(v["avg_us"], 2) for k, v in kern.items()},
            "decodes_enqueued_behind_their_scan": chained[0], "decoded_again_with_the_count": chained[1],
            "note": "frame_offsets = NULL: device record-boundary scan + decode. The decode is enqueued behind the scan, grids sized by a bound, "
                    "the frame count read on the device (no host wait between scan and decode). value = ASYNC (scan on its own stream beside "
                    "the decode of the batch before); sync_value = one finished batch per call (scan and decode back to back on one stream); "
                    "kernels_us from a second chain with the library's per-kernel events on (slower than the timed one)"}


def _gen_segment(args):
    """One ~1 GiB segment of the cfg4 stream (worker process): the cfg3 generator with its own seed and LSN base."""
    import numpy as np

    from etl_amd import synth
    k, nbytes = args
    w = synth.Workload([synth.table_mixed()], 0xE710004 + k, rows_per_txn=500, mix=(60, 30, 10), upd_key=10, upd_toast=5,
                       start_lsn=0x1000000 + (k << 36), name="cfg4_segment")
    parts, got = [], 0
    while got < nbytes:
        b, _ = w.fill(min(256 << 20, nbytes - got + (1 << 20)))
        if len(b) == 0:
            break
        parts.append(b)
        got += len(b)
    return np.concatenate(parts)


def leg_cfg4(dev_id, dev, world, rank, total_gib, seg_mib, gather_arenas, dist):
    """BASELINE configs[3]: ONE stream of `total_gib` GiB (8 segments of the cfg3 generator, consecutive LSN ranges) cut into
    `world` commit-aligned shards. Per rank, timed: device record-boundary scan of the shard, device frame tags, control-frame
    broadcast, decode in <= 1 GiB batches, header all-gather (+ the padded arena all-gather with --gather-arenas)."""
    import multiprocessing as mp

    import numpy as np
    import torch

    from etl_amd import abi, shard, synth
    from etl_amd.decoder import Decoder
    nseg = max(world, (total_gib << 10) // seg_mib)
    mine = [k for k in range(nseg) if k * world // nseg == rank]    # contiguous segments of this rank
    t_gen = time.perf_counter()
    with mp.get_context("fork").Pool(min(len(mine), max(1, (os.cpu_count() or 8) // max(world, 1)))) as pool:
        segs = pool.map(_gen_segment, [(k, seg_mib << 20) for k in mine])
    t_gen = time.perf_counter() - t_gen
    w = synth.Workload([synth.table_mixed()], 0xE710004, name="cfg4")
    dec = Decoder(dev_id)
    w.register(dec)
    d_segs = [torch.from_numpy(s).to(dev) for s in segs]
    nbytes = sum(len(s) for s in segs)
    cap_frames = max(len(s) for s in segs) // 24 + 1024
    d_offs = [torch.empty(cap_frames + 2, dtype=torch.int32, device=dev) for _ in segs]
    d_tags = [torch.empty(cap_frames + 2, dtype=torch.uint8, device=dev) for _ in segs]
    torch.cuda.synchronize()

    def run(check):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frames, keep, ctrl = 0, [], []
        for s, o, t in zip(d_segs, d_offs, d_tags):   # (1) boundaries + control stream on the device: per segment the host reads back a
                                                      #     frame count and 8 bytes (control-frame count, last tag) — no segment bytes, no tags
            nf = dec.scan_boundaries_device(s.data_ptr(), s.numel(), o.data_ptr(), o.numel())
            cb, co, last = dec.control_stream(s.data_ptr(), s.numel(), o.data_ptr(), nf)
            assert last == ord("C"), "a segment ends after a Commit"
            if len(co) > 1:   # rare: the transactions that hold Relation / DDL frames, reduced to {Begin, control frames, Commit}
                ctrl.append((cb, co))
            frames += nf
            keep.append((s, o, nf))
        if dist is not None:   # (2) control frames of earlier ranks (cfg4 has none after the schemas are primed: an empty exchange)
            mine_ctrl = (np.concatenate([c[0] for c in ctrl]) if ctrl else np.zeros(0, np.uint8), np.zeros(1, np.uint32))
            shard.replay_control(dec, [x for x in shard.all_gather_control(mine_ctrl)[:rank] if len(x[1]) > 1])
        dec.reset_stream_state()
        batches = []
        fl = abi.F_OUTPUT_ON_DEVICE | abi.F_ASYNC | (abi.F_NO_CONTROL if not ctrl else 0)   # the assertion only when step (1) found no control frame
        hdrs = torch.zeros((len(keep), 8), dtype=torch.int64, device=dev)
        for i, (s, o, nf) in enumerate(keep):          # (3) decode, one batch per segment (<= 1 GiB)
            b = dec.decode_device(s.data_ptr(), s.numel(), o.data_ptr(), nf, fl)
            b.header_to_device(hdrs[i].data_ptr())
            batches.append((b, nf))
        lay = None
        if dist is not None:                           # (4) one all-gather of the 64-byte headers: the global layout
            g = torch.empty((world, hdrs.numel()), dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(g, hdrs.reshape(1, -1))
        ev = ob = 0
        for b, nf in batches:
            rc = b.sync()
            v = b.view()
            assert not check or (rc == 0 and v.n_frames == nf), (rc, b.error)
            ev += v.n_events
            ob += HEADER_BYTES_PER_EVENT * v.n_events + v.fixed_bytes + v.heap_bytes
        gathered = 0
        if dist is not None and gather_arenas:         # (5) optional: the arenas themselves, padded, to every rank
            import ctypes as C
            for b, nf in batches:
                v = b.view()
                arrs = {}
                for name, ptr, n, item in (("kind", v.ev_kind, v.n_events, 1), ("flags", v.ev_flags, v.n_events, 1),
                                           ("table_id", v.ev_table_id, v.n_events, 4), ("schema_slot", v.ev_schema_slot, v.n_events, 4),
                                           ("start_lsn", v.ev_start_lsn, v.n_events, 8), ("commit_lsn", v.ev_commit_lsn, v.n_events, 8),
                                           ("tx_ordinal", v.ev_tx_ordinal, v.n_events, 8), ("body_off", v.ev_body_off, v.n_events, 8),
                                           ("fixed", v.fixed, v.fixed_bytes, 1), ("heap", v.heap, v.heap_bytes, 1)):
                    arrs[name] = abi.device_tensor(C.cast(ptr, C.c_void_p).value or 0, int(n) * item, dev)
                g2, lens = shard.all_gather_arenas(arrs)
                gathered += sum(int(x.numel()) for x in g2.values())
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        for b, _ in batches:
            b.close()
        return dt, frames, ev, ob, gathered

    run(True)
    dt, frames, ev, ob, gathered = min((run(True) for _ in range(2)), key=lambda r: r[0])
    tot = torch.tensor([nbytes, frames, ev], dtype=torch.int64, device=dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tot)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dec.profile(True)
    run(False)
    kern = kernel_table(dec.profile_read())
    dec.profile(False)
    dec.close()
    dt = float(tmax.item())
    return {"value": round(int(tot[0]) / dt / 1e9, 3), "unit": "GB/s", "events_per_s": round(int(tot[2]) / dt, 1), "n_gpus": world,
            "hbm_read_frac": round(int(tot[0]) / dt / 1e9 / (HBM_PEAK_GBPS * world), 5), "seconds": round(dt, 4),
            "stream_bytes": int(tot[0]), "frames": int(tot[1]), "shard_bytes_rank0": nbytes, "segments_per_rank": len(mine),
            "generation_s_rank0": round(t_gen, 1), "arena_bytes_gathered_rank0": gathered,
            "kernels_us_rank0": {k: round(v["avg_us"], 1) for k, v in kern.items()},
            "workload": f"cfg4: one {total_gib} GiB stream (cfg3 generator, {nseg} segments of {seg_mib} MiB with consecutive LSN ranges) cut into "
                        f"{world} commit-aligned shard(s); per rank: device boundary scan + frame tags, control broadcast, decode in "
                        f"{seg_mib} MiB batches, header all-gather" + (", padded arena all-gather" if gather_arenas else "")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--inner", type=int, default=1000, help="64 MiB batches decoded per step (20 steps of 1000 batches: a timed region of about a second)")
    ap.add_argument("--batch-mib", type=int, default=64)
    ap.add_argument("--pool", type=int, default=6, help="distinct batches resident in HBM (rotated)")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg4"])
    ap.add_argument("--legs", default="cfg1,cfg3,cfg5,wide70,copy,no_sidecar,handoff,default_flags,cfg2_mixed,pcie,cfg4", help="extra legs on rank 0 (comma separated; empty = none)")
    ap.add_argument("--cfg4-leg", action="store_true", help="add the cfg4 leg to a cfg2 / cfg3 run")
    ap.add_argument("--cfg4-gib", type=int, default=8)
    ap.add_argument("--cfg4-seg-mib", type=int, default=1024)
    ap.add_argument("--gather-arenas", action="store_true", help="cfg4: also all-gather the arenas (padded) to every rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the all-cores CPU leg (0 = the cores available, at most 64; 1 = skip)")
    ap.add_argument("--cpu-threads-seconds", type=float, default=6.0)
    ap.add_argument("--no-check", action="store_true", help="profiling ablations only")
    ap.add_argument("--no-scan-leg", action="store_true")
    ap.add_argument("--prime", type=int, default=WINDOW + 2, help="decodes that grow the arena pool before warm-up (profiling runs under rocprofv3 --pmc use 2)")
    ap.add_argument("--gather-every", type=int, default=4, help="N > 1: batches per header all-gather (their 64-byte headers travel together)")
    args = ap.parse_args()
    legs = [x for x in args.legs.split(",") if x]
    if args.no_scan_leg and "no_sidecar" in legs:
        legs.remove("no_sidecar")

    import torch
    import torch.distributed as dist

    from etl_amd import abi, shard, synth
    from etl_amd.decoder import Decoder

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ETLG_BENCH_FORCE_GATHER=1 (with torch.distributed.run --nproc-per-node 1) runs the collective path on one GPU
    gather = world > 1 or (os.environ.get("ETLG_BENCH_FORCE_GATHER") == "1" and "RANK" in os.environ)
    if gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, (world, args.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    check = not args.no_check

    if args.workload == "cfg4":
        r = leg_cfg4(local_rank, dev, world, rank, args.cfg4_gib, args.cfg4_seg_mib, args.gather_arenas, dist if gather else None)
        if rank == 0:
            out = {"metric": "WAL bytes/s decoded", "value": r["value"], "unit": "GB/s", "n_gpus": world, "steps": 1, "warmup": 1,
                   "ms_per_step": round(1000.0 * r["seconds"], 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                   "dtype": "u8", "data": "synthetic", "events_per_s": r["events_per_s"], "hbm_read_frac": r["hbm_read_frac"],
                   "config": {"workload": r["workload"], "parallelism": f"shard{world}"}, "cfg4": r}
            print(json.dumps(out))
        if gather:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- headline workload: this rank's shard = its own contiguous range of the stream
    mk = synth.cfg2 if args.workload == "cfg2" else synth.cfg3
    w = mk()
    cap = args.batch_mib << 20
    # every rank walks the same deterministic stream and keeps batches rank, rank+world, ...
    # (contiguous commit-aligned ranges; rank order == LSN order inside each step)
    pool = []
    i = 0
    while len(pool) < args.pool:
        buf, offs = w.fill(cap)
        if i % world == rank:
            pool.append((buf, offs))
        i += 1
    items = to_device(pool, dev)

    dec = Decoder(local_rank)
    stream = torch.cuda.Stream(device=dev)   # decode kernels, header copy and the all-gather share one stream
    torch.cuda.set_stream(stream)
    dec.set_stream(stream.cuda_stream)
    w.register(dec)
    # One 64-byte header slot per batch; with N > 1 the headers of G consecutive batches travel in ONE
    # asynchronous all-gather that overlaps the following decodes (etl_amd/shard.py: HeaderGatherer).
    G = max(1, args.gather_every)
    inner = max(1, args.inner)
    nwarm = ((args.warmup * inner + G - 1) // G) * G
    nbatches = nwarm + args.steps * inner
    hg = shard.HeaderGatherer(nbatches, G, dev, world=world, fence=dec.fence) if gather else None
    hdrs = hg.headers if gather else torch.zeros((nbatches + G, 8), dtype=torch.int64, device=dev)
    flags = abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL | abi.F_ASYNC

    # Arena pool priming (setup, like allocating buffers): the timed loop keeps WINDOW batches in flight, each holding its
    # own output arena; the library grows that pool on demand with hipMalloc, which serialises host and device.
    p = Pipeline(dec, items, flags, check)
    for _ in range(args.prime):
        p.issue()
    p.drain()
    p = Pipeline(dec, items, flags, check)
    for k in range(nwarm):
        p.issue(hdrs[k].data_ptr())
        if gather:
            hg.batch_done(k)
    p.drain()
    if gather:
        hg.flush(nwarm)
        hg.wait()
    torch.cuda.synchronize()

    # ---- timed region: exactly K steps (K x inner batches), barrier + synchronize on both sides
    p = Pipeline(dec, items, flags, check)
    if gather:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps * inner):
        p.issue(hdrs[nwarm + k].data_ptr())
        if gather:
            hg.batch_done(nwarm + k)
    if gather:
        hg.flush(nbatches)
    p.drain()
    ta = time.perf_counter()
    if gather:
        hg.wait()
    torch.cuda.synchronize()
    if gather:
        dist.barrier()
    t1 = time.perf_counter()
    if os.environ.get("ETLG_BENCH_DEBUG"):
        print(f"[bench debug] enqueue+drain {1e3 * (ta - t0):.2f} ms, gathers+sync+barrier {1e3 * (t1 - ta):.2f} ms", file=sys.stderr)
    elapsed = t1 - t0
    my_bytes, my_frames = p.bytes, p.frames
    if gather:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([my_bytes, my_frames], dtype=torch.int64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        all_bytes, all_frames = int(tot[0].item()), int(tot[1].item())
        lay = shard.global_layout(hg.headers_of(nbatches - 1).cpu().numpy())   # the last batch's headers, rank-major
        assert args.no_check or not lay["any_error"]
    else:
        all_bytes, all_frames = my_bytes, my_frames

    # ---- roofline leg: HIP-event timing of every kernel over a fresh run of 4 x inner batches (rank 0)
    roof = None
    paths = None
    if rank == 0:
        # Consecutive ASYNC batches run side by side on the library's two decode streams, so the per-launch HIP-event durations of
        # the chain overlap (their sum exceeds the wall clock). `roofline` is therefore priced on the kernel timed ALONE — the same
        # chain kept on one stream (profile mode 2: what `ETLG_OVERLAP=0 rocprofv3 --kernel-trace --stats` shows, profiles/) — and
        # carries beside it the overlapped per-launch duration and the launch interval of the timed region above.
        nprof = min(max(4 * inner, 20), 400) if args.prime > 2 else 4
        dec.profile(2)
        q = Pipeline(dec, items, flags, check)
        for _ in range(nprof):
            q.issue()
        q.drain()
        torch.cuda.synchronize()
        kern = kernel_table(dec.profile_read())
        dec.profile(False)
        for _try in range(3):   # (see leg_workload: a glitched event pair is measured again)
            dec.profile(True)
            q2 = Pipeline(dec, items, flags, check)
            for _ in range(nprof):
                q2.issue()
            q2.drain()
            torch.cuda.synchronize()
            kern2 = kernel_table(dec.profile_read())
            dec.profile(False)
            if all(kern2[k]["avg_us"] < 20 * kern[k]["avg_us"] + 1000 for k in kern2 if k in kern):
                break
        alg_bytes = (q.bytes + SIDECAR_BYTES_PER_FRAME * q.frames + q.out_bytes) / nprof
        # HBM traffic per launch comes from separate rocprofv3 --pmc passes over this same command
        # (tools/traffic.sh writes the file; FETCH_SIZE doubled per MI355X_MICROARCH.md, gfx950 note)
        tf = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
        dom = max(kern, key=lambda k: kern[k]["avg_us"] * kern[k]["launches"])
        traffic = traffic_of(tf, dom, args.batch_mib)
        roof = roofline_of(kern, alg_bytes, traffic, kern2)
        interval_us = 1e6 * elapsed / (args.steps * inner)
        roof["two_streams"] = {"batches_beside_their_predecessor": dec.debug_overlapped(),
                               "kernel_avg_us_overlapped": round(kern2[dom]["avg_us"], 2) if dom in kern2 else None,
                               "launch_interval_us": round(interval_us, 2),
                               "effective_GBps": round(alg_bytes / (interval_us * 1e-6) / 1e9, 1),
                               "effective_frac": round(alg_bytes / (interval_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)}
        paths = {**dec.debug_paths(), **dec.debug_rows()}

    extra = {}
    _leg_t = [time.perf_counter()]

    def _leg_done(name):   # wall time of every leg, to stderr (the JSON line stays the only stdout)
        t = time.perf_counter()
        print(f"[bench] {name}: {t - _leg_t[0]:.1f} s", file=sys.stderr, flush=True)
        _leg_t[0] = t
    _leg_done("headline + roofline leg")
    if rank == 0:
        extra["deferred_cells"] = deferred_cells(dec, items[0])
        if "no_sidecar" in legs and args.workload == "cfg2":
            extra["no_sidecar"] = leg_no_sidecar(local_rank, items, 160, check)
            _leg_done("no_sidecar")
        if "cfg3" in legs and args.workload != "cfg3":
            # (400 batches: a 60-batch region is 8 ms, and one scheduling hiccup of the host moved the figure by a third from call to call)
            extra["cfg3"] = leg_async(synth.cfg3, local_rank, dev, cap, 4, 400, flags, check,
                                      os.path.join(ROOT, "profiles", "traffic_cfg3.json"))[0]
            _leg_done("cfg3")
        if "default_flags" in legs and args.workload == "cfg2":
            # the headline workload WITHOUT the caller's no-control assertion: the optimistic path (first kernel as if there were
            # no Relation / DDL frame, ETLG_E_CTRL_HINT otherwise), ASYNC chain as with the assertion
            d = leg_async(synth.cfg2, local_rank, dev, cap, 6, 200, abi.F_OUTPUT_ON_DEVICE | abi.F_ASYNC, check)[0]   # the headline's pool and batch count
            extra["default_flags"] = {k: d[k] for k in ("value", "unit", "workload", "batches", "paths")}
            extra["default_flags"]["kernels_us"] = d["roofline"]["pipeline_kernels_us"]
            _leg_done("default_flags")
        if "cfg2_mixed" in legs and args.workload == "cfg2":
            extra["cfg2_mixed"] = leg_cfg2_mixed(local_rank, dev, cap, 10, 240)
            _leg_done("cfg2_mixed")
        if "cfg5" in legs:
            extra["cfg5"] = leg_cfg5(local_rank, dev, cap, 16, 2)
            _leg_done("cfg5")
        if "cfg1" in legs:
            extra["cfg1"] = leg_cfg1(local_rank, dev, cap, 4.0)
            _leg_done("cfg1")
        if "wide70" in legs:
            extra["wide70"] = leg_wide70(local_rank, dev, 44000, 40)   # (~64 MiB per batch, like the other legs; 24 000 rows = 36 MB until round 5)
            _leg_done("wide70")
        if "copy" in legs:
            extra["copy"] = leg_copy(local_rank, dev, 400000, 8)
            _leg_done("copy")
        if "handoff" in legs:
            extra["handoff"] = leg_handoff(local_rank, dev, cap, 5)
            _leg_done("handoff")
        if "pcie" in legs:
            extra["pcie"] = leg_pcie(local_rank, dev, cap, 24)
            _leg_done("pcie")
    if (args.cfg4_leg or "cfg4" in legs) and world == 1 and args.workload != "cfg4":
        extra["cfg4"] = leg_cfg4(local_rank, dev, 1, 0, args.cfg4_gib if args.cfg4_leg else 2, args.cfg4_seg_mib, False, None)   # the default line: 2 GiB of the 64 GiB stream on this one GPU
        _leg_done("cfg4")

    # ---- CPU baseline leg (rank 0, N == 1 only): the oracle on the same host cores
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        o = oracle.Oracle(mode=oracle.MODE_FULL)
        w.register(o)
        sample_bytes = sample_frames = 0
        secs = 0.0
        reps = 0
        while secs < args.cpu_seconds:
            buf, offs = pool[reps % len(pool)]
            o.reset_stream_state()
            s, ne, nf, ec = o.decode_timed(buf, offs)
            assert ec == 0 and nf == len(offs) - 1
            secs += s
            sample_bytes += len(buf)
            sample_frames += nf
            reps += 1
        cpu = {"value": round(sample_bytes / secs / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
               "events_per_s": round(sample_frames / secs, 1),
               "sample": f"{reps} x {args.batch_mib} MiB batches of {w.name} ({sample_bytes} bytes, {secs:.1f} s), "
                         "single thread = the reference's one apply task; C++ restatement of the Rust decoder "
                         "(oracle/, FULL mode: decode into an event object model, then drop it)"}
        nthr = cpu_threads(args.cpu_threads)
        if nthr > 1:
            cpu["all_cores"] = cpu_baseline_threads(w, pool, nthr, args.cpu_threads_seconds)
        _leg_done("cpu_baseline")

    if rank == 0:
        value = all_bytes / elapsed / 1e9
        out = {
            "metric": "WAL bytes/s decoded", "value": round(value, 3), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "events_per_s": round(all_frames / elapsed, 1),
            "hbm_read_frac": round(value / (HBM_PEAK_GBPS * world), 5),   # input bytes/s per GPU over the 8 TB/s spec
            "config": {"workload": f"{w.name}: {args.batch_mib} MiB batches of CopyData-framed pgoutput, "
                                   "device-resident in / device-resident out, offsets sidecar, NO_CONTROL | ASYNC",
                       "batches_per_step": inner, "batch_bytes": int(my_bytes / (args.steps * inner)),
                       "frames_per_batch": int(my_frames / (args.steps * inner)),
                       "pool_batches": len(pool), "parallelism": f"shard{world}", "kernel_paths": paths},
            "roofline": roof, "cpu_baseline": cpu,
        }
        out.update(extra)
        print(json.dumps(out))
    dec.close()
    if gather:
        dist.barrier()   # rank 0 runs the extra legs; everybody leaves together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
