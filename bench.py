#!/usr/bin/env python
"""bench.py — WAL bytes/s decoded on MI355X (BASELINE.json metric).

A "step" is one pass of the decode hot path over one 64 MiB batch of synthetic
WAL (BASELINE.json configs[1]: fixed-width 5 x int4 INSERT tuples, 113-byte
CopyData frames) that is already resident in HBM when the timed region
starts; the decoded arena stays in HBM. Batches rotate through a pool larger
than the 256 MiB Infinity Cache so steps do not re-read cached input.

  python bench.py [--gpus N --steps K --warmup W]

N > 1 is launched by the driver under torch.distributed.run (one rank per
GPU). Each rank decodes its own contiguous, commit-aligned shard of the
stream (weak scaling: one batch per rank per step); the only collective is an
all-gather of a 64-byte header per rank per step (RCCL over xGMI) that gives
every rank the LSN-ordered global layout. Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Algorithmic bytes of one launch: every input byte read once (frames + the 4-byte
# offsets sidecar per frame) and every arena byte written once (42 bytes of event header columns per
# event + the fixed row arena + the heap). cfg2: 113 + 4 + 42 + 24 = 183 bytes per row.
SIDECAR_BYTES_PER_FRAME = 4
HEADER_BYTES_PER_EVENT = 42     # kind 1, flags 1, table 4, slot 4, start 8, commit 8, ordinal 8, body offset 8


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-mib", type=int, default=64)
    ap.add_argument("--pool", type=int, default=6, help="distinct batches resident in HBM (rotated)")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the all-cores CPU leg (0 = the cores available, at most 64; 1 = skip)")
    ap.add_argument("--cpu-threads-seconds", type=float, default=6.0)
    ap.add_argument("--no-check", action="store_true", help="profiling ablations only")
    ap.add_argument("--no-scan-leg", action="store_true")
    ap.add_argument("--gather-every", type=int, default=4, help="N > 1: batches per header all-gather (their 64-byte headers travel together)")
    args = ap.parse_args()

    import numpy as np
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

    # ---- workload: this rank's shard = its own contiguous range of the stream
    mk = synth.cfg2 if args.workload == "cfg2" else synth.cfg3
    w = mk()
    cap = args.batch_mib << 20
    # every rank walks the same deterministic stream and keeps batches rank, rank+world, ...
    # (contiguous commit-aligned ranges; rank order == LSN order inside each step)
    pool = []
    need = args.pool
    i = 0
    while len(pool) < need:
        buf, offs = w.fill(cap)
        if i % world == rank:
            pool.append((buf, offs))
        i += 1
    d_in = [(torch.from_numpy(b).to(dev), torch.from_numpy(o.view(np.int32)).to(dev), len(b), len(o) - 1) for b, o in pool]
    torch.cuda.synchronize()

    dec = Decoder(local_rank)
    stream = torch.cuda.Stream(device=dev)   # decode kernels, header copy and the all-gather share one stream
    torch.cuda.set_stream(stream)
    dec.set_stream(stream.cuda_stream)
    w.register(dec)
    # One 64-byte header slot per step; with N > 1 the headers of G consecutive batches travel in ONE
    # asynchronous all-gather that overlaps the following decodes (etl_amd/shard.py: HeaderGatherer).
    G = max(1, args.gather_every)
    nbatches = ((args.warmup + G - 1) // G) * G + args.steps
    hg = shard.HeaderGatherer(nbatches, G, dev, world=world) if gather else None
    hdrs = hg.headers if gather else torch.zeros((nbatches + G, 8), dtype=torch.int64, device=dev)
    flags = abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL | abi.F_ASYNC

    def step(k, keep):
        tb, to, nbytes, nfr = d_in[k % len(d_in)]
        b = dec.decode_device(tb.data_ptr(), nbytes, to.data_ptr(), nfr, flags)
        b.header_to_device(hdrs[k].data_ptr())
        keep.append((b, nbytes, nfr, k))
        if gather:
            hg.batch_done(k)

    def flush_gathers(k_end):
        if gather:
            hg.flush(k_end)

    def wait_gathers():
        if gather:
            hg.wait()

    out_bytes = [0]   # arena bytes written by the drained batches (headers + fixed + heap)

    def drain(keep, check):
        tot_b = tot_f = 0
        for b, nbytes, nfr, g in keep:
            rc = b.sync()
            v = b.view()
            if check and not args.no_check:
                assert rc == 0 and v.n_events == nfr and v.n_frames == nfr, (rc, v.n_events, v.n_frames, nfr, b.error)
            out_bytes[0] += HEADER_BYTES_PER_EVENT * v.n_events + v.fixed_bytes + v.heap_bytes
            tot_b += nbytes
            tot_f += nfr
            b.close()
        return tot_b, tot_f

    # Arena pool priming (setup, like allocating buffers): the timed loop keeps `steps` batches in flight, each
    # holding its own output arena; the library grows that pool on demand with hipMalloc, which serialises host
    # and device, so the pool is grown to its working size here, before warm-up, and only reused afterwards.
    prime = [dec.decode_device(d_in[k % len(d_in)][0].data_ptr(), d_in[k % len(d_in)][2], d_in[k % len(d_in)][1].data_ptr(),
                               d_in[k % len(d_in)][3], flags) for k in range(args.steps + 1)]
    torch.cuda.synchronize()
    for b in prime:
        b.sync(); b.close()
    del prime

    keep = []
    for k in range(args.warmup):
        step(k, keep)
    flush_gathers(args.warmup)
    wait_gathers()
    torch.cuda.synchronize()
    drain(keep, True)

    # ---- timed region: exactly K steps, barrier + synchronize on both sides
    keep = []
    if gather:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k0 = ((args.warmup + G - 1) // G) * G   # the timed steps start on a group boundary
    for k in range(args.steps):
        step(k0 + k, keep)
    flush_gathers(k0 + args.steps)
    ta = time.perf_counter()
    wait_gathers()
    tb_ = time.perf_counter()
    torch.cuda.synchronize()
    tc = time.perf_counter()
    if gather:
        dist.barrier()
    t1 = time.perf_counter()
    if os.environ.get("ETLG_BENCH_DEBUG"):
        print(f"[bench debug] enqueue {1e3 * (ta - t0):.2f} ms, wait gathers {1e3 * (tb_ - ta):.2f} ms, sync {1e3 * (tc - tb_):.2f} ms, "
              f"barrier {1e3 * (t1 - tc):.2f} ms", file=sys.stderr)
    elapsed = t1 - t0
    last_k = keep[-1][3]
    my_bytes, my_frames = drain(keep, True)
    if gather:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([my_bytes, my_frames], dtype=torch.int64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        all_bytes, all_frames = int(tot[0].item()), int(tot[1].item())
        lay = shard.global_layout(hg.headers_of(last_k).cpu().numpy())   # the last batch's headers, rank-major
        assert args.no_check or not lay["any_error"]
    else:
        all_bytes, all_frames = my_bytes, my_frames

    # ---- roofline leg: HIP-event timing of every kernel over a fresh run of K steps (rank 0)
    roof = None
    kern = {}
    if rank == 0:
        dec.profile(True)
        keep = []
        for k in range(args.steps):
            tb, to, nbytes, nfr = d_in[k % len(d_in)]
            keep.append((dec.decode_device(tb.data_ptr(), nbytes, to.data_ptr(), nfr, flags), nbytes, nfr, None))
        torch.cuda.synchronize()
        prof = dec.profile_read()
        dec.profile(False)
        out_bytes[0] = 0
        pb, pf = drain(keep, True)
        kern = {k: {"launches": n, "avg_us": 1000.0 * ms / n} for k, (n, ms) in prof.items() if n}
        dom = max(kern, key=lambda k: kern[k]["avg_us"])
        alg_bytes = (pb + SIDECAR_BYTES_PER_FRAME * pf + out_bytes[0]) / args.steps
        ach = alg_bytes / (kern[dom]["avg_us"] * 1e-6) / 1e9
        pipe_us = sum(v["avg_us"] for v in kern.values())
        # HBM traffic per launch comes from separate rocprofv3 --pmc passes over this same command
        # (tools/traffic.sh writes the file; FETCH_SIZE doubled per MI355X_MICROARCH.md, gfx950 note)
        traffic = None
        tf = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
        if os.path.exists(tf):
            t = json.load(open(tf))
            if t.get("kernel") == dom and t.get("batch_mib") == args.batch_mib:
                traffic = t["hbm_bytes_per_launch"]
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "alg_bytes_per_launch": int(alg_bytes), "kernel_avg_us": round(kern[dom]["avg_us"], 2),
                "pipeline_kernels_us": {k: round(v["avg_us"], 2) for k, v in kern.items()},
                "pipeline_sum_us": round(pipe_us, 2),
                "pipeline_frac": round(alg_bytes / (pipe_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)}

    # ---- no-sidecar leg (rank 0): the same batches with frame_offsets = NULL, i.e. the record-boundary
    #      scan runs on the device first (scan.hip). Reported beside `value`, never as `value`: the
    #      reference's host learns every frame length from its socket codec, so a sidecar costs it nothing.
    scan = None
    if rank == 0 and not args.no_scan_leg:
        fl = abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL
        for k in range(2):
            tb, to, nbytes, nfr = d_in[k % len(d_in)]
            dec.decode_device(tb.data_ptr(), nbytes, None, 0, fl).close()
        dec.profile(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nb = 0
        for k in range(args.steps):
            tb, to, nbytes, nfr = d_in[k % len(d_in)]
            b = dec.decode_device(tb.data_ptr(), nbytes, None, 0, fl)
            assert args.no_check or (b.rc == 0 and b.view().n_frames == nfr)
            nb += nbytes
            b.close()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        prof = dec.profile_read()
        dec.profile(False)
        kb = prof.get("k_bounds", (0, 0.0))
        scan = {"value": round(nb / (t1 - t0) / 1e9, 3), "unit": "GB/s", "k_bounds_avg_us": round(1000.0 * kb[1] / max(kb[0], 1), 2),
                "k_bounds_launches_per_batch": round(kb[0] / args.steps, 2),
                "note": "frame_offsets = NULL: device record-boundary scan + decode, synchronous (the host reads the frame count back)"}

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
                                   "device-resident in / device-resident out, offsets sidecar, NO_CONTROL",
                       "batch_bytes": int(my_bytes / args.steps), "frames_per_batch": int(my_frames / args.steps),
                       "pool_batches": len(pool), "parallelism": f"shard{world}"},
            "roofline": roof, "cpu_baseline": cpu, "no_sidecar": scan,
        }
        print(json.dumps(out))
    dec.close()
    if gather:
        dist.barrier()   # rank 0 runs the extra legs; everybody leaves together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
