// Fused single-pass decode kernel for gfx950 (MI355X, wave64).
//
// One workgroup = one tile of BLK consecutive CopyData frames, one lane per frame:
//
//   1. the tile's bytes are staged into LDS with coalesced 16-byte loads (every
//      input byte leaves HBM exactly once; all parsing then reads LDS),
//   2. each lane classifies its frame, walks the tuples (field-length decode) and
//      contributes to a workgroup scan of the transaction state,
//   3. the tile aggregate is published and the running prefix over earlier tiles
//      is obtained by a decoupled look-back (single pass, no kernel boundary):
//      transaction state first, then (events, fixed bytes, heap bytes),
//   4. lanes decode their cells and write events / rows / heap entries at their
//      final, compacted positions.
//
// The look-back words carry status + payload in ONE 64-bit word written by a
// single relaxed agent-scope atomic store and read by relaxed agent-scope
// atomic loads, so no fence is needed and nothing depends on dispatch order or
// XCD placement (tile ids come from an atomic ticket, so every predecessor of a
// tile has already started; spins are bounded).
//
// The fused kernel only produces a result when the whole batch decodes without
// error: the first-error cut (exact totals at the failing frame) is computed by
// the multi-pass kernels in kernels.hip, which the host falls back to when this
// kernel reports any error. Errors end the stream in the reference (fail-fast,
// crates/etl/src/replication/apply.rs:2475-2481), so that path is cold.
#include "lookback.hip.h"

#ifndef ETLG_MINWAVES
#define ETLG_MINWAVES 4     // waves per SIMD the register allocator must leave room for (128 VGPRs; measured 108 vs 119 us at 3)
#endif
#ifndef ETLG_LB_PARALLEL
#define ETLG_LB_PARALLEL 1 // run the three independent look-backs on three waves
#endif
#include "fixed_tile.hip.h"  // fixed-width plan: schema-constant sizing for tiles of Begin / Commit / fixed-width Insert frames

namespace etlg {

// Everything after staging; `base` points at byte `win0` of the input (LDS or global).
// `p`: parameters whose side-table pointers may point into LDS; `pg`: the original (global) ones.
template <int BLK, bool STAGED>
DEV void tile_body(const DecParams& p, const DecParams& pg, const FusedParams& q, uint32_t tile, uint32_t nt,
                   const uint32_t* s_offs, const u8* base, uint32_t win0, uint32_t* s32, uint64_t* s64) {
  const uint32_t tid = threadIdx.x;
  const int wave = tid >> 6;
  const bool live = tid < nt;
  const uint32_t f = tile * BLK + tid;
  uint32_t* fail = (q.dbg & 4) ? nullptr : &p.res->fused_fail;
  // ---- phase 1: envelope, tag, structure (field-length decode of every tuple)
  FrameView v{f, 0, base, base};
  RowMsg m;
  bool wire_ok = true;
  uint32_t o0 = 0;
  if (live) {
    o0 = s_offs[tid];
    const uint32_t o1 = s_offs[tid + 1];
    if (o1 > o0 && o1 <= p.in_len) {
      v.fr = base + (o0 - win0);
      v.e = base + (o1 - win0);
      v.tag = classify_ptr(v.fr, o1 - o0);
    }
    wire_ok = frame_structure(v, m, STAGED);
  }
  TSTAMP(2);
  uint32_t cnt = 0, mark = 0;
  if (live) {
    if (consumes_ordinal(v.tag)) cnt = 1;
    if (v.tag == 'B') { cnt |= 0x80000000u; mark = ((o0 + 1) << 1) | 1; }
    if (v.tag == 'C') mark = (o0 + 1) << 1;
  }
  uint32_t seg_in, pm, tot_cnt, tot_mark;
  block_scan_txn(cnt, mark, s32, seg_in, pm, tot_cnt, tot_mark);
  TSTAMP(3);
  const uint64_t txn_agg = ((uint64_t)seg_pack30(tot_cnt) << 32) | tot_mark;
  TxnCtx tx{true, 0, 0};
  uint32_t bc = 0, bm = 0;
  auto make_tx = [&]() {
    bc = s32[12]; bm = s32[13];
    const uint32_t seg = seg_combine(bc, seg_in);
    const uint32_t last = bm > pm ? bm : pm;
    tx.in_txn = (last & 1u) != 0;
    // the Begin that opened this frame's transaction: carried in from earlier tiles (its final_lsn was
    // fetched by the wave that ran the transaction look-back) or a frame of this tile (read in place)
    tx.final_lsn = !tx.in_txn ? 0 : last == bm ? s64[6] : ld_be64(base + (((last >> 1) - 1) - win0) + kBodyOff);
    const uint64_t c = seg & 0x7FFFFFFFu;
    tx.ord = (seg & 0x80000000u) ? c - 1 : s64[3] + c - 1;
  };
  if (q.seq_lookback) {
    // ownership depends on the transaction's final_lsn (a table is SyncDone): transaction state first
    if (wave == 0) txn_lookback(p, q.d_txn, q.ntiles, tile, txn_agg, fail, s32, s64);
    __syncthreads();
    make_tx();
  }
  TSTAMP(4);
  // ---- phase 2: exact output sizes
  uint32_t emit = 0, fixed = 0, heap = 0;
  uint64_t pay[3] = {0, 0, 0};
  int row_slot = -1;
  if (live) size_frame(p, v, tx, wire_ok, m, emit, fixed, heap, pay, row_slot, q.seq_lookback != 0, STAGED);
  TSTAMP(5);
  uint32_t x_ev = emit, x_fx = fixed >> 2, x_hp = heap >> 2, tot3[3];
  block_scan3_excl(x_ev, x_fx, x_hp, s32, tot3);
  {  // payload byte counters (A3): workgroup reduce through LDS, then one atomic per tile and
     // counter into a shard (a single hot address would serialise thousands of atomics in L2)
    // (a batch is < 2 GiB on this path, so every partial sum fits 32 bits)
    const uint32_t a0 = wave_last(wave_scan_add((uint32_t)pay[0])), a1 = wave_last(wave_scan_add((uint32_t)pay[1])),
                   a2 = wave_last(wave_scan_add((uint32_t)pay[2]));
    if ((tid & 63) == 0) {  // s64[0..3] is free here: block_scan3_excl only used s32
      if (a0) atomicAdd((unsigned long long*)&s64[0], (unsigned long long)a0);
      if (a1) atomicAdd((unsigned long long*)&s64[1], (unsigned long long)a1);
      if (a2) atomicAdd((unsigned long long*)&s64[2], (unsigned long long)a2);
    }
  }
  TSTAMP(6);
  // ---- look-back: output positions (and the transaction state when it was not needed earlier);
  //      independent prefixes run on different waves so their latencies overlap
  const uint64_t agg_a = ((uint64_t)tot3[0] << 32) | tot3[2];
  const uint64_t agg_b = tot3[1];
  if (BLK >= 192 && ETLG_LB_PARALLEL) {
    if (wave == 0) { const uint64_t a = lookback<OpAdd2>(q.d_outa, q.d_outa + q.ntiles, tile, agg_a, 0, fail); if (tid == 0) s64[4] = a; }
    if (wave == 1) { const uint64_t b = lookback<OpAdd>(q.d_outb, q.d_outb + q.ntiles, tile, agg_b, 0, fail); if ((tid & 63) == 0) s64[5] = b; }
    if (wave == 2 && !q.seq_lookback) txn_lookback(p, q.d_txn, q.ntiles, tile, txn_agg, fail, s32, s64);
  } else if (wave == 0) {
    const uint64_t a = lookback<OpAdd2>(q.d_outa, q.d_outa + q.ntiles, tile, agg_a, 0, fail);
    const uint64_t b = lookback<OpAdd>(q.d_outb, q.d_outb + q.ntiles, tile, agg_b, 0, fail);
    if (tid == 0) { s64[4] = a; s64[5] = b; }
    if (!q.seq_lookback) txn_lookback(p, q.d_txn, q.ntiles, tile, txn_agg, fail, s32, s64);
  }
  __syncthreads();
  if (tid < 3 && s64[tid]) atomicAdd(&p.res->pay_shard[tile & 31][tid], (unsigned long long)s64[tid]);
  TSTAMP(7);
  if (!q.seq_lookback) {
    make_tx();
    if (live && wire_ok) txn_check_frame(p, v, tx);
  }
  const uint64_t pre_ev = s64[4] >> 32, pre_hp = (uint64_t)(uint32_t)s64[4] << 2, pre_fx = s64[5] << 2;
  if (tid == 0 && tile == q.ntiles - 1) {  // the last tile knows the totals and the carried transaction state
    DevResult* r = p.res;
    r->n_events = pre_ev + tot3[0]; r->fixed_bytes = pre_fx + ((uint64_t)tot3[1] << 2); r->heap_bytes = pre_hp + ((uint64_t)tot3[2] << 2);
    r->n_frames = p.nframes;
    const uint32_t sg = seg_combine(bc, tot_cnt);
    const uint32_t lm = bm > tot_mark ? bm : tot_mark;
    const bool it = (lm & 1u) != 0;
    r->out_in_txn = it;
    r->out_final_lsn = it ? (lm == bm ? s64[6] : final_lsn_of_mark(p, lm)) : 0;
    const uint64_t c = sg & 0x7FFFFFFFu;
    r->out_next_ord = (sg & 0x80000000u) ? c : s64[3] + c;
    carry_publish(r);
  }
  // ---- phase 3: decode + write
  if (!emit) return;
  const uint64_t ev_idx = pre_ev + x_ev, fx_off = pre_fx + ((uint64_t)x_fx << 2), hp_off = pre_hp + ((uint64_t)x_hp << 2);
  if (fx_off + fixed > p.fixed_cap || hp_off + heap > p.heap_cap || hp_off + heap > 0xFFFFFFFFull) {
    record_error(p, f, RK_DECODE, ETLG_E_WIRE);
    return;
  }
  if (q.dbg & 2) return;
  write_frame(p, v, tx, m, row_slot, ev_idx, fx_off, hp_off, &pg, STAGED);
  TSTAMP(8);
}

template <int BLK>
__global__ __launch_bounds__(BLK, ETLG_MINWAVES) void k_fused(DecParams pg, FusedParams q) {
  ETLG_DYNAMIC_LDS(smem);
  __shared__ uint32_t s_offs[BLK + 1];
  __shared__ uint32_t s32[16];
  __shared__ uint64_t s64[8];
  const uint32_t tid = threadIdx.x;
  if (q.clear_words) {  // descriptors are double buffered: this launch clears the buffer the next batch will use
    const uint32_t per = (q.clear_words + gridDim.x - 1) / gridDim.x;
    for (uint32_t i = tid; i < per; i += BLK) { const uint32_t w = blockIdx.x * per + i; if (w < q.clear_words) q.d_clear[w] = 0; }
  }
  if (!(pg.flags & 16u) && !load_carry(pg)) return;  // ASYNC chain: the state the batch before this one left (flags bit 4: read late, by the tiles that need it — txn_lookback)
  DecParams p = pg;  // side-table pointers of `p` are redirected to the LDS copy below
  if ((q.dbg & 8) && tid == 0) s64[7] = clock64();
  if (tid < 3) s64[tid] = 0;  // per-tile payload accumulators
  // Kernel head with two dependent global round trips: the tile id is blockIdx.x (nothing to broadcast through LDS first); the
  // four side tables are read as ONE concatenation, up to four dwords per lane, and only stored to LDS after the staging loads
  // have been issued; the span comes through real scalar loads (constant address space) that overlap the side-table loads; the
  // tile's bytes are one more round trip (eight 16-byte loads in flight per lane). The barrier after staging covers the side
  // tables too. (Round 1 measured this head, the fixed-width plan and the register / address-space fixes together: 78.2 us
  // against 87.8 us for the head with one round trip per table, profiles/r01j_ab_quick_*.json.)
  const uint32_t tile = blockIdx.x;
  if (tile >= q.ntiles) return;
  const uint32_t f0 = tile * BLK;
  const uint32_t nt = p.nframes - f0 < (uint32_t)BLK ? p.nframes - f0 : (uint32_t)BLK;
  const uint32_t f0u = __builtin_amdgcn_readfirstlane(f0), ntu = __builtin_amdgcn_readfirstlane(nt);
  SideRegs side;
  side_load<BLK>(p, q.side_bytes != 0, (uint32_t*)smem, tid, side);
  const ETLG_CONST_AS uint32_t* offs_c = (const ETLG_CONST_AS uint32_t*)(uintptr_t)p.offs;
  const uint32_t span0 = offs_c[f0u], span1 = offs_c[f0u + ntu];
  const uint32_t my_o = tid <= nt ? p.offs[f0 + tid] : 0u;
  const uint32_t last_o = (tid == 0 && nt == (uint32_t)BLK) ? p.offs[f0 + BLK] : 0u;
  u8* stage = smem + q.side_bytes;
  const uint32_t a0 = span0 & ~15u;
  const bool window_ok = q.in_aligned && span1 > span0 && span1 <= p.in_len && (uint64_t)(span1 - a0) + 16 <= q.lds_bytes - q.side_bytes && !(q.dbg & 1);
  if (window_ok) {
    // coalesced staging: 16 B per lane per step; the tail that would cross in_len goes bytewise
    const uint32_t full_end = a0 + ((span1 - a0) & ~15u);  // last full 16-byte chunk boundary <= span1
    // eight independent 16-byte loads in flight per lane before the first LDS store: one HBM round trip for a 29 KB tile
    stage_chunks<BLK>(p.in, stage, a0, full_end, tid);
    for (uint32_t c = full_end + tid; c < span1; c += BLK) stage[c - a0] = p.in[c];
  }
  side_store<BLK>((uint32_t*)smem, tid, side);
  if (tid <= nt) s_offs[tid] = my_o;
  if (tid == 0 && nt == (uint32_t)BLK) s_offs[BLK] = last_o;
  __syncthreads();
  TSTAMP(0);
  // every well-formed frame of the tile must lie inside [span0, span1) for the staged copy to be used
  bool lane_ok = true;
  if (tid < nt) {
    const uint32_t o0 = s_offs[tid], o1 = s_offs[tid + 1];
    lane_ok = o1 <= o0 || o1 > p.in_len || (o0 >= span0 && o1 <= span1);
  }
  const bool use_lds = __syncthreads_and(lane_ok ? 1 : 0) && window_ok;
  if (use_lds && q.side_bytes && !q.seq_lookback && (q.dbg & ~64u) == 0 && !(pg.flags & 0xF02u) && pg.worker_kind == ETLG_WORKER_APPLY) {
    const uint32_t nt4 = pg.n_tables * (sizeof(DevTable) / 4), ne4 = pg.n_epochs * (sizeof(DevEpoch) / 4), ns4 = pg.n_slots * (sizeof(DevSlot) / 4);
    DecParams pl = pg;  // side tables through pointers that can only name LDS
    uint32_t* b0 = (uint32_t*)smem;
    pl.tables = (const DevTable*)b0; pl.epochs = (const DevEpoch*)(b0 + nt4);
    pl.slots = (const DevSlot*)(b0 + nt4 + ne4); pl.cols = (const DevCol*)(b0 + nt4 + ne4 + ns4);
    if (tile_fixed<BLK>(pl, pg, q, tile, nt, s_offs, stage, a0, s32, s64)) {
      if ((q.dbg & 64u) && tid == 0) atomicAdd(&pg.res->dbg_t[11], 1ull);  // ETLG_FUSED_DBG=64: tiles that took the fixed-width plan (tests)
      return;
    }
  }
  // the staged instance reads its side tables through pointers that can only name LDS (ds_read instead of flat
  // loads that must first find out which memory they address); a tile without them takes the global instance
  if (use_lds && q.side_bytes) {
    TSTAMP(1);
    const uint32_t nt4 = pg.n_tables * (sizeof(DevTable) / 4), ne4 = pg.n_epochs * (sizeof(DevEpoch) / 4), ns4 = pg.n_slots * (sizeof(DevSlot) / 4);
    DecParams pl = pg;
    uint32_t* b0 = (uint32_t*)smem;
    pl.tables = (const DevTable*)b0; pl.epochs = (const DevEpoch*)(b0 + nt4);
    pl.slots = (const DevSlot*)(b0 + nt4 + ne4); pl.cols = (const DevCol*)(b0 + nt4 + ne4 + ns4);
    tile_body<BLK, true>(pl, pg, q, tile, nt, s_offs, stage, a0, s32, s64);
  } else {
    tile_body<BLK, false>(p, pg, q, tile, nt, s_offs, p.in, 0, s32, s64);
  }
}

}  // namespace etlg

extern "C" {

using namespace etlg;

// blk: 256 or 64 frames per tile. lds_bytes = dynamic LDS per workgroup.
void etlg_k_launch_fused(int blk, const DecParams* p, const void* qv, hipStream_t s) {
  const FusedParams* q = (const FusedParams*)qv;
  if (blk == 256) hipLaunchKernelGGL(k_fused<256>, dim3(q->ntiles), dim3(256), q->lds_bytes, s, *p, *q);
  else hipLaunchKernelGGL(k_fused<64>, dim3(q->ntiles), dim3(64), q->lds_bytes, s, *p, *q);
}

int etlg_k_fused_set_lds(void) {
  // allow the full 160 KiB of LDS as dynamic shared memory
  hipError_t e1 = hipFuncSetAttribute((const void*)k_fused<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
  hipError_t e2 = hipFuncSetAttribute((const void*)k_fused<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
  return (e1 == hipSuccess && e2 == hipSuccess) ? 0 : 1;
}

}  // extern "C"
