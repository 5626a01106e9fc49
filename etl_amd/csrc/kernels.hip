// gfx950 (MI355X, CDNA4, wave64) kernels of the batched pgoutput decode stage.
//
// Pipeline for one batch of N CopyData frames (one lane per frame, kBlock
// frames per workgroup):
//
//   k_classify   frame envelope check, pgoutput tag, per-block aggregates of the
//                transaction scan (A0/A1: ReplicationMessage::parse,
//                reference call sites crates/etl/src/replication/apply.rs:2037-2075)
//   k_scan_txn   exclusive scan of the per-block aggregates (single workgroup)
//   k_ctrl_list  compaction of R / M frames for the host control plane (A14)
//   k_size       transaction state per frame (A2: apply.rs:2279-2617, 942-963),
//                ownership (apply.rs:2836-2867), structural validation of the
//                whole message, exact output sizes, payload byte accounting
//                (A3: codec/event.rs:261-297)
//   k_scan_out   exclusive scan of per-block output sizes
//   k_write      tuple -> row (A4/A5: codec/event.rs:554-587, 938-983), typed
//                text -> value (A6-A10: codec/text.rs:32-153, bool.rs, hex.rs,
//                time.rs, etl-postgres numeric.rs/time.rs), update/delete
//                assembly (A11/A12: codec/event.rs:437-527, 605-923),
//                begin/commit/truncate (A13), into the arena of include/etlg.h
//   k_finalize   first-error cut, totals, carried transaction state
//
// Integer / byte work only: no MFMA. All inter-workgroup data flows through
// kernel boundaries (no in-kernel hand-offs), so there are no spin waits.
#include "codec.hip.h"

namespace etlg {

// ------------------------------------------------------------ k_classify
__global__ __launch_bounds__(kBlock) void k_classify(DecParams p) {
  __shared__ uint32_t lds[8];
  const uint32_t f = blockIdx.x * kBlock + threadIdx.x;
  uint32_t tag = 0, cnt = 0, mark = 0;
  if (f < p.nframes) {
    tag = classify_frame(p, f);
    p.f_tag[f] = (u8)tag;
    if (consumes_ordinal(tag)) cnt = 1;
    if (tag == 'B') { cnt |= 0x80000000u; mark = ((f + 1) << 1) | 1; }
    if (tag == 'C') mark = (f + 1) << 1;
  }
  uint32_t tot_cnt, tot_mark;
  block_scan_incl<2>(cnt, lds, &tot_cnt);
  block_scan_incl<1>(mark, lds + 4, &tot_mark);
  if (threadIdx.x == 0) { p.blk_cnt[blockIdx.x] = tot_cnt; p.blk_last[blockIdx.x] = tot_mark; }
}

// Single workgroup: exclusive scans over per-block aggregates (sequential
// chunks of kBlock blocks).
__global__ __launch_bounds__(kBlock) void k_scan_txn(DecParams p) {
  __shared__ uint32_t lds[8];
  uint32_t run_cnt = 0;                 // (flag=0, count=0): identity for seg_combine
  // carried state: virtual Begin before frame 0. flags bit 5 (the pre-pass of a pipelined control batch, host.cpp ctl_begin): the
  // state is whatever the batch before this one left in ITS block (`carry`: a pre-pass block or a decode result block, complete by
  // stream order), and this pre-pass leaves the state at its own end in `res` for the next one
  const bool chained = (p.flags & 32u) && p.carry;
  const uint32_t in0 = chained ? p.carry->out_in_txn : p.in_txn;
  uint32_t run_mark = in0 ? 1u : 0u;
  for (uint32_t base = 0; base < p.nblocks; base += kBlock) {
    const uint32_t b = base + threadIdx.x;
    uint32_t c = b < p.nblocks ? p.blk_cnt[b] : 0, m = b < p.nblocks ? p.blk_last[b] : 0;
    uint32_t tc, tm;
    uint32_t ic = block_scan_incl<2>(c, lds, &tc);
    uint32_t im = block_scan_incl<1>(m, lds + 4, &tm);
    // exclusive = running ⊕ (inclusive of previous lane)
    uint32_t pc = __shfl_up(ic, 1, 64), pm = __shfl_up(im, 1, 64);
    __shared__ uint32_t edge[8];
    if ((threadIdx.x & 63) == 63) { edge[threadIdx.x >> 6] = ic; edge[4 + (threadIdx.x >> 6)] = im; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && threadIdx.x) { pc = edge[(threadIdx.x >> 6) - 1]; pm = edge[4 + (threadIdx.x >> 6) - 1]; }
    uint32_t ec = threadIdx.x ? seg_combine(run_cnt, pc) : run_cnt;
    uint32_t em = threadIdx.x ? (run_mark > pm ? run_mark : pm) : run_mark;
    if (b < p.nblocks) { p.blk_cnt[b] = ec; p.blk_last[b] = em; }
    run_cnt = seg_combine(run_cnt, tc);
    run_mark = run_mark > tm ? run_mark : tm;
    __syncthreads();
  }
  if (threadIdx.x == 0) {  // totals at [nblocks]
    p.blk_cnt[p.nblocks] = run_cnt;
    p.blk_last[p.nblocks] = run_mark;
    if (p.flags & 32u) {
      const uint64_t lsn0 = chained ? p.carry->out_final_lsn : p.final_lsn;
      p.res->out_in_txn = run_mark & 1u;
      p.res->out_final_lsn = !(run_mark & 1u) ? 0ull : run_mark == 1u ? lsn0 : ld_be64(p.in + p.offs[(run_mark >> 1) - 1] + kBodyOff);
    }
  }
}

// Per-frame transaction context from the block-local scans + block prefixes.
DEV TxnCtx txn_context(const DecParams& p, uint32_t f, uint32_t tag, uint32_t* lds) {
  uint32_t cnt = 0, mark = 0;
  if (f < p.nframes) {
    if (consumes_ordinal(tag)) cnt = 1;
    if (tag == 'B') { cnt |= 0x80000000u; mark = ((f + 1) << 1) | 1; }
    if (tag == 'C') mark = (f + 1) << 1;
  }
  uint32_t ic = block_scan_incl<2>(cnt, lds, nullptr);
  uint32_t im = block_scan_incl<1>(mark, lds + 4, nullptr);
  // exclusive max = max over lanes before this one
  uint32_t pm = __shfl_up(im, 1, 64);
  __shared__ uint32_t edge[4];
  if ((threadIdx.x & 63) == 63) edge[threadIdx.x >> 6] = im;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) pm = threadIdx.x ? edge[(threadIdx.x >> 6) - 1] : 0;
  __syncthreads();
  const uint32_t bc = p.blk_cnt[blockIdx.x], bm = p.blk_last[blockIdx.x];
  const uint32_t seg = seg_combine(bc, ic);
  const uint32_t last = bm > pm ? bm : pm;  // last B/C strictly before f
  TxnCtx t;
  t.in_txn = (last & 1u) != 0;
  t.final_lsn = 0;
  if (t.in_txn) {
    if (last == 1u) t.final_lsn = ((p.flags & 32u) && p.carry) ? p.carry->out_final_lsn : p.final_lsn;  // Begin came in a previous batch
    else t.final_lsn = ld_be64(p.in + p.offs[(last >> 1) - 1] + kBodyOff);
  }
  const uint64_t c = seg & 0x7FFFFFFFu;
  t.ord = (seg & 0x80000000u) ? c - 1 : p.next_ord + c - 1;
  return t;
}

// ------------------------------------------------------------ k_ctrl_list
__global__ __launch_bounds__(kBlock) void k_ctrl_list(DecParams p) {
  __shared__ uint32_t lds[8];
  const uint32_t f = blockIdx.x * kBlock + threadIdx.x;
  const uint32_t tag = f < p.nframes ? p.f_tag[f] : 0;
  TxnCtx t = txn_context(p, f, tag, lds);
  // the frames' bytes go to one staging buffer (the host used to fetch every Relation / DDL frame with a copy of its own,
  // ~10 us each: a stream with a DDL every 200 transactions spent 1.3 ms per 64 MiB batch there); the whole block copies
  __shared__ uint32_t n_here;
  __shared__ uint32_t job[kBlock][3];  // (input offset, length, staging offset) of the block's control frames
  if (threadIdx.x == 0) n_here = 0;
  __syncthreads();
  if (tag == 'R' || tag == 'M') {
    uint32_t i = atomicAdd(&p.res->n_ctrl, 1u);
    if (i < p.ctrl_cap) {
      CtrlFrame c;
      c.frame = f; c.tag = tag; c.in_txn = t.in_txn; c.stage_off = 0xFFFFFFFFu; c.final_lsn = t.final_lsn;
      c.o0 = p.offs[f]; c.o1 = p.offs[f + 1];
      const uint32_t len = c.o1 - c.o0;
      if (p.ctrl_stage && len <= p.ctrl_stage_cap) {
        const uint32_t pos = atomicAdd(&p.res->ctrl_bytes, (len + 3u) & ~3u);
        if (pos <= p.ctrl_stage_cap - len) {
          c.stage_off = pos;
          const uint32_t j = atomicAdd(&n_here, 1u);
          job[j][0] = c.o0; job[j][1] = len; job[j][2] = pos;
        }
      }
      p.ctrl[i] = c;
    }
  }
  __syncthreads();
  const uint32_t nj = n_here;
  for (uint32_t j = 0; j < nj; j++) {
    const uint32_t o0 = job[j][0], len = job[j][1], pos = job[j][2];
    for (uint32_t k = threadIdx.x; k < len; k += kBlock) p.ctrl_stage[pos + k] = p.in[o0 + k];
  }
}

__global__ __launch_bounds__(kBlock) void k_size(DecParams p) {
  __shared__ uint32_t lds[8];
  __shared__ uint64_t lds64[4];
  const uint32_t f = blockIdx.x * kBlock + threadIdx.x;
  const bool live = f < p.nframes && f < p.host_err_frame;
  const uint32_t tag = f < p.nframes ? p.f_tag[f] : 0;
  const TxnCtx tx = txn_context(p, f, tag, lds);
  uint32_t emit = 0, fixed = 0, heap = 0;
  uint64_t pay[3] = {0, 0, 0};
  if (live) {
    FrameView v{f, tag, p.in + p.offs[f], p.in + p.offs[f + 1]};
    RowMsg m;
    const bool ok = frame_structure(v, m);
    int row_slot = -1;
    size_frame(p, v, tx, ok, m, emit, fixed, heap, pay, row_slot);
  }
  if (f < p.nframes) { p.f_emit[f] = (u8)emit; p.f_fixed[f] = fixed; p.f_heap[f] = heap; }
  uint32_t tot_ev;
  block_scan_incl<0>(emit, lds, &tot_ev);
  const uint64_t tf = block_sum64(fixed, lds64);
  const uint64_t th = block_sum64(heap, lds64);
  const uint64_t p0 = block_sum64(pay[0], lds64), p1 = block_sum64(pay[1], lds64), p2 = block_sum64(pay[2], lds64);
  if (threadIdx.x == 0) {
    p.blk_ev[blockIdx.x] = tot_ev; p.blk_fixed[blockIdx.x] = tf; p.blk_heap[blockIdx.x] = th;
    p.blk_payload[3 * blockIdx.x + 0] = p0; p.blk_payload[3 * blockIdx.x + 1] = p1; p.blk_payload[3 * blockIdx.x + 2] = p2;
  }
}

// Exclusive scans of per-block output sizes (single workgroup).
__global__ __launch_bounds__(kBlock) void k_scan_out(DecParams p) {
  __shared__ uint64_t lds64[4];
  uint64_t run_ev = 0, run_fx = 0, run_hp = 0;
  for (uint32_t base = 0; base < p.nblocks; base += kBlock) {
    const uint32_t b = base + threadIdx.x;
    const bool in = b < p.nblocks;
    uint64_t ev = in ? p.blk_ev[b] : 0, fx = in ? p.blk_fixed[b] : 0, hp = in ? p.blk_heap[b] : 0;
    uint64_t tev, tfx, thp;
    uint64_t xev = block_scan_excl64(ev, lds64, &tev);
    uint64_t xfx = block_scan_excl64(fx, lds64, &tfx);
    uint64_t xhp = block_scan_excl64(hp, lds64, &thp);
    if (in) { p.blk_ev[b] = (uint32_t)(run_ev + xev); p.blk_fixed[b] = run_fx + xfx; p.blk_heap[b] = run_hp + xhp; }
    run_ev += tev; run_fx += tfx; run_hp += thp;
  }
  if (threadIdx.x == 0) { p.blk_ev[p.nblocks] = (uint32_t)run_ev; p.blk_fixed[p.nblocks] = run_fx; p.blk_heap[p.nblocks] = run_hp; }
}

__global__ __launch_bounds__(kBlock) void k_write(DecParams p) {
  __shared__ uint32_t lds[8];
  __shared__ uint64_t lds64[4];
  const uint32_t f = blockIdx.x * kBlock + threadIdx.x;
  const bool in_range = f < p.nframes;
  const uint32_t tag = in_range ? p.f_tag[f] : 0;
  const TxnCtx tx = txn_context(p, f, tag, lds);
  const uint32_t emit = in_range ? p.f_emit[f] : 0;
  const uint64_t fx = in_range ? p.f_fixed[f] : 0, hp = in_range ? p.f_heap[f] : 0;
  const uint64_t ev_idx = p.blk_ev[blockIdx.x] + block_scan_excl64(emit, lds64, nullptr);
  const uint64_t fx_off = p.blk_fixed[blockIdx.x] + block_scan_excl64(fx, lds64, nullptr);
  const uint64_t hp_off = p.blk_heap[blockIdx.x] + block_scan_excl64(hp, lds64, nullptr);
  if (!emit) return;
  if (fx_off + fx > p.fixed_cap || hp_off + hp > p.heap_cap || hp_off + hp > 0xFFFFFFFFull) {
    record_error(p, f, RK_DECODE, ETLG_E_WIRE);  // capacity guard (cannot happen with the host's bounds)
    return;
  }
  FrameView v{f, tag, p.in + p.offs[f], p.in + p.offs[f + 1]};
  RowMsg m;
  (void)frame_structure(v, m);
  write_frame(p, v, tx, m, -1, ev_idx, fx_off, hp_off);
}

// --------------------------------------------------------------- k_finalize
// Single workgroup. Cuts the batch at the first failing frame, computes the
// totals and the transaction state carried into the next batch.
__global__ __launch_bounds__(kBlock) void k_finalize(DecParams p) {
  __shared__ uint32_t lds[8];
  __shared__ uint64_t lds64[4];
  DevResult* r = p.res;
  unsigned long long fe = r->first_err;
  uint32_t stop = p.nframes < p.host_err_frame ? p.nframes : p.host_err_frame;  // frames consumed
  uint32_t err_rank = 0xFF;
  if (fe != kNoErr) {
    const uint32_t ef = (uint32_t)(fe >> 16);
    if (ef < stop) { stop = ef; err_rank = (uint32_t)((fe >> 8) & 0xFF); }
    else if (ef == stop) err_rank = (uint32_t)((fe >> 8) & 0xFF);
  }
  // totals at `stop`: block prefix + in-block partial sums
  const uint32_t sb = stop / kBlock;  // block holding the stop frame (== nblocks when stop == nframes and aligned)
  uint64_t ev = 0, fx = 0, hp = 0, pay[3] = {0, 0, 0};
  if (sb >= p.nblocks) { ev = p.blk_ev[p.nblocks]; fx = p.blk_fixed[p.nblocks]; hp = p.blk_heap[p.nblocks]; }
  else { ev = p.blk_ev[sb]; fx = p.blk_fixed[sb]; hp = p.blk_heap[sb]; }
  // payload: whole blocks before sb
  uint64_t a0 = 0, a1 = 0, a2 = 0;
  for (uint32_t b = threadIdx.x; b < sb && b < p.nblocks; b += kBlock) { a0 += p.blk_payload[3 * b]; a1 += p.blk_payload[3 * b + 1]; a2 += p.blk_payload[3 * b + 2]; }
  pay[0] = block_sum64(a0, lds64); pay[1] = block_sum64(a1, lds64); pay[2] = block_sum64(a2, lds64);
  // partial block: frames [sb*kBlock, stop) (+ the failing frame's payload when the
  // error was raised after the metrics were recorded: apply.rs:2459-2463)
  const uint32_t f = sb * kBlock + threadIdx.x;
  uint64_t e1 = 0, f1 = 0, h1 = 0, q0 = 0, q1 = 0, q2 = 0;
  if (sb < p.nblocks && f < p.nframes) {
    const bool before = f < stop;
    const bool at_err = f == stop && err_rank != 0xFF && err_rank >= RK_SCHEMA;
    if (before) { e1 = p.f_emit[f]; f1 = p.f_fixed[f]; h1 = p.f_heap[f]; }
    if (before || at_err) {
      const uint32_t tag = p.f_tag[f];
      if (tag == 'I' || tag == 'U' || tag == 'D') {
        RowMsg m;
        if (parse_row_msg(tag, p.in + p.offs[f] + kBodyOff, p.in + p.offs[f + 1], m)) {
          // only frames inside a transaction record metrics (others fail first)
          if (tag == 'I') q0 = m.vbytes; else if (tag == 'U') q1 = m.vbytes; else q2 = m.vbytes;
        }
      }
    }
  }
  ev += block_sum64(e1, lds64); fx += block_sum64(f1, lds64); hp += block_sum64(h1, lds64);
  pay[0] += block_sum64(q0, lds64); pay[1] += block_sum64(q1, lds64); pay[2] += block_sum64(q2, lds64);
  // carried transaction state after the last consumed frame = context of frame `stop`
  // computed as if it were a non-consuming frame.
  uint32_t cnt = 0, mark = 0;
  if (sb < p.nblocks && f < stop) {
    const uint32_t tag = p.f_tag[f];
    if (consumes_ordinal(tag)) cnt = 1;
    if (tag == 'B') { cnt |= 0x80000000u; mark = ((f + 1) << 1) | 1; }
    if (tag == 'C') mark = (f + 1) << 1;
  }
  uint32_t tc, tm;
  block_scan_incl<2>(cnt, lds, &tc);
  block_scan_incl<1>(mark, lds + 4, &tm);
  if (threadIdx.x == 0) {
    const uint32_t bc = sb < p.nblocks ? p.blk_cnt[sb] : p.blk_cnt[p.nblocks];
    const uint32_t bm = sb < p.nblocks ? p.blk_last[sb] : p.blk_last[p.nblocks];
    const uint32_t seg = sb < p.nblocks ? seg_combine(bc, tc) : bc;
    uint32_t last = sb < p.nblocks ? (bm > tm ? bm : tm) : bm;
    // a failed Commit has already taken remote_final_lsn (apply.rs:2303) before the LSN check
    bool in_txn = (last & 1u) != 0;
    uint64_t fin = 0;
    if (in_txn) fin = last == 1u ? p.final_lsn : ld_be64(p.in + p.offs[(last >> 1) - 1] + kBodyOff);
    if (fe != kNoErr && (uint32_t)(fe >> 16) == stop && (uint32_t)(fe & 0xFF) == ETLG_E_COMMIT_LSN) in_txn = false;
    const uint64_t c = seg & 0x7FFFFFFFu;
    uint64_t next = (seg & 0x80000000u) ? c : p.next_ord + c;
    // the failing frame consumed its ordinal if it got past the transaction-state check
    if (err_rank != 0xFF && err_rank >= RK_SCHEMA && stop < p.nframes && consumes_ordinal(p.f_tag[stop])) next += 1;
    r->n_events = ev; r->fixed_bytes = fx; r->heap_bytes = hp;
    r->payload[0] = pay[0]; r->payload[1] = pay[1]; r->payload[2] = pay[2];
    r->n_frames = stop;
    r->out_in_txn = in_txn; r->out_final_lsn = fin; r->out_next_ord = next;
  }
}

// ------------------------------------------------------------ control stream of a frame range (multi-GPU recipe)
// Every shard broadcasts the transactions of its range that hold a Relation ('R') or logical-decoding Message ('M') frame,
// reduced to {Begin, those frames, Commit}: apply.rs:2160-2276 and 2363-2440 are the only writers of the schema store and the
// shared table cache. Three small kernels over the tags k_classify left: pick the control frames (rare: usually none, and the
// host stops after the count), find each one's Begin / Commit, gather the kept frames' bytes.
__global__ __launch_bounds__(kBlock) void k_ctl_pick(const u8* tags, uint32_t nframes, uint32_t* hdr /* [0] count, [1] last tag */, uint32_t* list, uint32_t cap) {
  const uint32_t f = blockIdx.x * kBlock + threadIdx.x;
  if (f >= nframes) return;
  const uint32_t t = tags[f];
  if (f == nframes - 1) hdr[1] = t;
  if (t == 'R' || t == 'M') { const uint32_t i = atomicAdd(&hdr[0], 1u); if (i < cap) list[i] = f; }
}
// span[2 i] = frame of the Begin whose transaction holds control frame list[i], span[2 i + 1] = frame of that transaction's Commit
// (0xFFFFFFFF: none — the frame stands outside a transaction of this range, or the transaction is still open at its end)
__global__ __launch_bounds__(kBlock) void k_ctl_span(const u8* tags, uint32_t nframes, const uint32_t* list, uint32_t n, uint32_t* span) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const uint32_t f = list[i];
  uint32_t b = 0xFFFFFFFFu, c = 0xFFFFFFFFu;
  for (uint32_t k = f; k-- > 0;) { const uint32_t t = tags[k]; if (t == 'B') { b = k; break; } if (t == 'C') break; }
  if (b != 0xFFFFFFFFu) for (uint32_t k = f + 1; k < nframes; k++) if (tags[k] == 'C') { c = k; break; }
  span[2 * i] = b; span[2 * i + 1] = c;
}
// one workgroup per kept frame: job[3 j] = input offset, [3 j + 1] = length, [3 j + 2] = output offset
__global__ __launch_bounds__(kBlock) void k_ctl_gather(const u8* in, const uint32_t* offs, const uint32_t* frames, uint32_t* lens, const uint32_t* out_offs, u8* out) {
  const uint32_t f = frames[blockIdx.x];
  const uint32_t o0 = offs[f], len = offs[f + 1] - o0;
  if (!out) { if (threadIdx.x == 0) lens[blockIdx.x] = len; return; }
  const uint32_t pos = out_offs[blockIdx.x];
  for (uint32_t k = threadIdx.x; k < len; k += kBlock) out[pos + k] = in[o0 + k];
}

// ------------------------------------------------------------------ launch
}  // namespace etlg

extern "C" {

using namespace etlg;

// which: 0 classify, 1 scan_txn, 2 ctrl_list, 3 size, 4 scan_out, 5 write, 6 finalize
void etlg_k_launch(int which, const DecParams* p, hipStream_t s) {
  switch (which) {
    case 0: hipLaunchKernelGGL(k_classify, dim3(p->nblocks), dim3(kBlock), 0, s, *p); break;
    case 1: hipLaunchKernelGGL(k_scan_txn, dim3(1), dim3(kBlock), 0, s, *p); break;
    case 2: hipLaunchKernelGGL(k_ctrl_list, dim3(p->nblocks), dim3(kBlock), 0, s, *p); break;
    case 3: hipLaunchKernelGGL(k_size, dim3(p->nblocks), dim3(kBlock), 0, s, *p); break;
    case 4: hipLaunchKernelGGL(k_scan_out, dim3(1), dim3(kBlock), 0, s, *p); break;
    case 5: hipLaunchKernelGGL(k_write, dim3(p->nblocks), dim3(kBlock), 0, s, *p); break;
    case 6: hipLaunchKernelGGL(k_finalize, dim3(1), dim3(kBlock), 0, s, *p); break;
    default: break;
  }
}

// Commit-aligned shard cuts (etlg_shard_plan): WAVE k finds the k-th interior cut of an n-way split balanced by bytes — the first frame
// index e with tags[e - 1] == 'C' whose byte offset reaches total * k / n, or the last such index when no Commit lies behind that point
// (0 when the range has no Commit at all). The caller makes the list monotonic. The wave looks at 64 tags per step (ballot): a stream
// with few or no Commits costs nframes / 64 steps per cut instead of one lane walking every tag.
__global__ __launch_bounds__(64) void k_shard_cuts(const u8* tags, const uint32_t* offs, uint32_t nframes, uint32_t n_shards, uint32_t* cuts) {
  const uint32_t k = blockIdx.x + 1, lane = threadIdx.x;
  if (k >= n_shards) return;
  const uint64_t total = offs[nframes];
  const uint64_t target = total * k / n_shards;
  uint32_t lo = 0, hi = nframes;   // smallest i in [0, nframes] with offs[i] >= target
  while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if ((uint64_t)offs[mid] >= target) hi = mid; else lo = mid + 1; }
  uint32_t cut = 0;
  for (uint64_t e0 = lo ? lo : 1u; e0 <= nframes && !cut; e0 += 64) {   // forward: the first e >= max(lo, 1) with a Commit in front of it
    const uint64_t e = e0 + lane;
    const unsigned long long m = __ballot(e <= nframes && tags[e - 1] == 'C');
    if (m) cut = (uint32_t)(e0 + (uint32_t)__builtin_ctzll(m));
  }
  for (int64_t e1 = lo; e1 >= 1 && !cut; e1 -= 64) {                   // none behind the target: the last one at or before it
    const int64_t e = e1 - (int64_t)lane;
    const unsigned long long m = __ballot(e >= 1 && tags[e - 1] == 'C');
    if (m) cut = (uint32_t)(e1 - (int64_t)__builtin_ctzll(m));
  }
  if (lane == 0) cuts[k - 1] = cut;
}
void etlg_k_shard_cuts(const uint8_t* tags, const uint32_t* offs, uint32_t nframes, uint32_t n_shards, uint32_t* cuts, hipStream_t s) {
  if (n_shards > 1) hipLaunchKernelGGL(k_shard_cuts, dim3(n_shards - 1), dim3(64), 0, s, tags, offs, nframes, n_shards, cuts);
}

void etlg_k_ctl_pick(const uint8_t* tags, uint32_t nframes, uint32_t* hdr, uint32_t* list, uint32_t cap, hipStream_t s) {
  hipLaunchKernelGGL(k_ctl_pick, dim3((nframes + kBlock - 1) / kBlock), dim3(kBlock), 0, s, tags, nframes, hdr, list, cap);
}
void etlg_k_ctl_span(const uint8_t* tags, uint32_t nframes, const uint32_t* list, uint32_t n, uint32_t* span, hipStream_t s) {
  hipLaunchKernelGGL(k_ctl_span, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, tags, nframes, list, n, span);
}
void etlg_k_ctl_gather(const uint8_t* in, const uint32_t* offs, const uint32_t* frames, uint32_t nkeep, uint32_t* lens, const uint32_t* out_offs, uint8_t* out, hipStream_t s) {
  hipLaunchKernelGGL(k_ctl_gather, dim3(nkeep), dim3(kBlock), 0, s, in, offs, frames, lens, out_offs, out);
}

const char* etlg_k_name(int which) {
  static const char* names[] = {"k_classify", "k_scan_txn", "k_ctrl_list", "k_size", "k_scan_out", "k_write", "k_finalize"};
  return which >= 0 && which < 7 ? names[which] : "?";
}

}  // extern "C"
