// "Fixed-width plan" of the generic fused kernel. Whole batches of such frames go to k_plan (plan.hip, the lean kernel);
// this body serves the tiles of k_fused launches that conform anyway: tables with uuid columns (not in k_plan's class set),
// batches k_plan gave up because ONE tile did not conform, runs of a fixed-width table between runs of another one.
//
// Round 1 priced k_fused at 1 949 VALU instructions per wave of 64 rows on cfg2, 473 of them between the
// structure walk and the look-back: the generic sizing (`size_frame`: any tag, ownership by transaction LSN,
// cache epochs, heap bytes of either tuple image), three workgroup scans and three payload reductions — for
// frames whose sizes are constants of their schema. A tile takes this body instead when, after the structure
// walk and BEFORE anything is published, all of its lanes agree (one workgroup vote) that
//   * every frame is a well-formed Begin, Commit or Insert,
//   * every Insert names a table the apply worker owns outright (state Ready: ownership does not depend on the
//     transaction), whose cache entry is Ready for the whole batch (no Relation / DDL epoch), and
//   * that table's schema slot is fixed-width (no cell can reach the heap).
// Then every live frame emits exactly one event (the event prefix is the lane index), the heap prefix is zero,
// one scan (fixed-arena dwords) replaces three, and only the INSERT payload counter is reduced. Everything the
// tile publishes (the three look-back descriptors, the last tile's totals, the payload shards) has the same
// format as in `tile_body`, so conforming and generic tiles mix freely inside one launch; errors that only
// show up while decoding (integer syntax, NULL in a required column, tuple width) are recorded by the shared
// row writer exactly as in the generic body. Any other shape — another tag, a wire error, an unknown or
// not-yet-ready table, a var-len schema — fails the vote and the tile runs `tile_body` from the top.
#pragma once

namespace etlg {

// Exclusive sum scan of one u32 over the workgroup (one LDS exchange). lds: nwaves words.
DEV uint32_t block_scan1_excl(uint32_t a, uint32_t* lds, uint32_t& tot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
  const uint32_t ia = wave_scan_add(a);
  if (lane == 63) lds[wave] = ia;
  __syncthreads();
  uint32_t pa = 0, ta = 0;
  for (int w = 0; w < nw; w++) {
    const uint32_t xa = lds[w];
    if (w < wave) pa += xa;
    ta += xa;
  }
  __syncthreads();
  tot = ta;
  return pa + ia - a;
}

// Returns false (workgroup-uniform, nothing published, nothing written) when the tile does not conform.
// Caller guarantees: staged tile with the side tables in LDS (`p`), apply worker, no table in SyncDone state
// (!q.seq_lookback), no table-copy rows, no profiling ablation (q.dbg clear but for the counting bit 64, p.flags bits 8-11 clear).
template <int BLK>
DEV bool tile_fixed(const DecParams& p, const DecParams& pg, const FusedParams& q, uint32_t tile, uint32_t nt,
                    const uint32_t* s_offs, const u8* base, uint32_t win0, uint32_t* s32, uint64_t* s64) {
  const uint32_t tid = threadIdx.x;
  const int wave = tid >> 6;
  const bool live = tid < nt;
  const uint32_t f = tile * BLK + tid;
  uint32_t* fail = &p.res->fused_fail;
  // ---- phase 1: envelope, tag, structure; conformance
  FrameView v{f, 0, base, base};
  RowMsg m;
  m.rel_id = 0; m.old_t = nullptr; m.new_t = nullptr; m.old_kind = ETLG_OLD_NONE; m.old_n = m.new_n = 0; m.vbytes = 0;
  uint32_t o0 = 0, fixed = 0, cnt = 0, mark = 0;
  int row_slot = -1;
  bool conform = true;
  if (live) {
    o0 = s_offs[tid];
    const uint32_t o1 = s_offs[tid + 1];
    if (o1 > o0 && o1 <= p.in_len) {
      v.fr = base + (o0 - win0);
      v.e = base + (o1 - win0);
      v.tag = classify_ptr(v.fr, o1 - o0);
    }
    cnt = 1;
    if (v.tag == 'B') { cnt |= 0x80000000u; mark = ((o0 + 1) << 1) | 1; fixed = 8; }
    else if (v.tag == 'C') { mark = (o0 + 1) << 1; fixed = 16; }
    else if (v.tag == 'I') {
      conform = parse_row_msg('I', v.fr + kBodyOff, v.e, m, true);
      const int ti = conform ? find_table(p, m.rel_id) : -1;
      if (ti < 0) conform = false;
      else {
        const DevTable& t = p.tables[ti];
        if (t.state_kind != ETLG_TS_READY || t.init_kind != 2u || t.ep_begin != t.ep_end || t.init_slot < 0) conform = false;
        else {
          const DevSlot& s = p.slots[t.init_slot];
          if (s.has_var) conform = false;
          row_slot = t.init_slot;
          fixed = s.row_full;
        }
      }
    } else conform = false;
  }
  if (!__syncthreads_and(conform ? 1 : 0)) return false;
  // ---- transaction scan (as tile_body)
  uint32_t seg_in, pm, tot_cnt, tot_mark;
  block_scan_txn(cnt, mark, s32, seg_in, pm, tot_cnt, tot_mark);
  const uint64_t txn_agg = ((uint64_t)seg_pack30(tot_cnt) << 32) | tot_mark;
  // ---- sizes: one event per live frame, no heap; only the fixed-arena prefix needs a scan
  uint32_t tot_fx;
  const uint32_t x_fx = block_scan1_excl(fixed >> 2, s32, tot_fx);
  {  // INSERT payload bytes (A3); Begin / Commit lanes carry 0
    const uint32_t a0 = wave_last(wave_scan_add(m.vbytes));
    if ((tid & 63) == 0 && a0) atomicAdd((unsigned long long*)&s64[0], (unsigned long long)a0);
  }
  // ---- look-back: same three descriptors as tile_body, one per wave where the tile has three waves
  const uint64_t agg_a = (uint64_t)nt << 32;  // (events, heap dwords)
  const uint64_t agg_b = tot_fx;
  if (BLK >= 192 && ETLG_LB_PARALLEL) {
    if (wave == 0) { const uint64_t a = lookback<OpAdd2>(q.d_outa, q.d_outa + q.ntiles, tile, agg_a, 0, fail); if (tid == 0) s64[4] = a; }
    if (wave == 1) { const uint64_t b = lookback<OpAdd>(q.d_outb, q.d_outb + q.ntiles, tile, agg_b, 0, fail); if ((tid & 63) == 0) s64[5] = b; }
    if (wave == 2) txn_lookback(pg, q.d_txn, q.ntiles, tile, txn_agg, fail, s32, s64);
  } else if (wave == 0) {
    const uint64_t a = lookback<OpAdd2>(q.d_outa, q.d_outa + q.ntiles, tile, agg_a, 0, fail);
    const uint64_t b = lookback<OpAdd>(q.d_outb, q.d_outb + q.ntiles, tile, agg_b, 0, fail);
    txn_lookback(pg, q.d_txn, q.ntiles, tile, txn_agg, fail, s32, s64);
    if (tid == 0) { s64[4] = a; s64[5] = b; }
  }
  __syncthreads();
  if (tid == 0 && s64[0]) atomicAdd(&p.res->pay_shard[tile & 31][0], (unsigned long long)s64[0]);
  // ---- transaction context of every frame (make_tx of tile_body) and the state checks that were deferred
  const uint32_t bc = s32[12], bm = s32[13];
  TxnCtx tx{true, 0, 0};
  {
    const uint32_t seg = seg_combine(bc, seg_in);
    const uint32_t last = bm > pm ? bm : pm;
    tx.in_txn = (last & 1u) != 0;
    tx.final_lsn = !tx.in_txn ? 0 : last == bm ? s64[6] : ld_be64(base + (((last >> 1) - 1) - win0) + kBodyOff);
    const uint64_t c = seg & 0x7FFFFFFFu;
    tx.ord = (seg & 0x80000000u) ? c - 1 : s64[3] + c - 1;
  }
  if (live) txn_check_frame(p, v, tx);
  const uint64_t pre_ev = s64[4] >> 32, pre_hp = (uint64_t)(uint32_t)s64[4] << 2, pre_fx = s64[5] << 2;
  if (tid == 0 && tile == q.ntiles - 1) {  // the last tile knows the totals and the carried transaction state
    DevResult* r = p.res;
    r->n_events = pre_ev + nt; r->fixed_bytes = pre_fx + ((uint64_t)tot_fx << 2); r->heap_bytes = pre_hp;
    r->n_frames = p.nframes;
    const uint32_t sg = seg_combine(bc, tot_cnt);
    const uint32_t lm = bm > tot_mark ? bm : tot_mark;
    const bool it = (lm & 1u) != 0;
    r->out_in_txn = it;
    r->out_final_lsn = it ? (lm == bm ? s64[6] : final_lsn_of_mark(pg, lm)) : 0;
    const uint64_t c = sg & 0x7FFFFFFFu;
    r->out_next_ord = (sg & 0x80000000u) ? c : s64[3] + c;
    carry_publish(r);
  }
  // ---- decode + write (the shared row writer; conforming INSERT waves of one table take its wave-uniform path)
  if (!live) return true;
  const uint64_t ev_idx = pre_ev + tid, fx_off = pre_fx + ((uint64_t)x_fx << 2);
  if (fx_off + fixed > p.fixed_cap || pre_hp > p.heap_cap || pre_hp > 0xFFFFFFFFull) {
    record_error(p, f, RK_DECODE, ETLG_E_WIRE);
    return true;
  }
  write_frame(p, v, tx, m, row_slot, ev_idx, fx_off, pre_hp, &pg, true);
  return true;
}

}  // namespace etlg
