// k_plan — the fixed-width decode plan for gfx950 (MI355X, wave64): the hot path of BASELINE cfg2.
//
// A batch is eligible (host.cpp, "plan") when the apply worker owns every table outright (state Ready, cache entry
// Ready for the whole batch) and the eligible tables' schemas only hold integer / oid / bool columns. The kernel then
// decodes Begin / Commit / Insert frames and nothing else: one WAVE per tile of 64 consecutive frames, one lane per
// frame, no workgroup barrier anywhere.
//
//   1. the tile's bytes go HBM -> LDS with global_load_lds_dwordx4 (LDS-DMA: no staging VGPRs, 1 KiB per
//      wave-instruction, every input byte leaves HBM once); occupancy (5-8 waves per SIMD at <= 64 VGPRs and
//      ~8 KB of LDS per wave) overlaps one wave's load with the others' parsing,
//   2. every LDS access is an ALIGNED dword read + v_alignbyte: lanes sit at arbitrary byte offsets (113-byte
//      frames) and a misaligned ds_read costs 3.3x an aligned one on this chip (tools/ubench/lds_align.hip;
//      70 % of k_fused's LDS cycles were SQ_LDS_UNALIGNED_STALL),
//   3. sizes are schema constants, so ONE 64-bit look-back descriptor per tile carries everything a tile needs from
//      its predecessors: {last Begin/Commit mark : 30, fixed-arena dwords : 32}. Events are one per frame (the event
//      prefix is the frame index), the heap is empty, ordinals follow from frame indexes (every frame of a
//      conforming batch consumes one: apply.rs:2284-2292, 2339, 2457). When every planned table has the same row size
//      even that is known before a frame is read — a Begin is 51 bytes on the wire, a Commit 56, every other frame a
//      row — and a pre-pass over the offsets sidecar (k_plan_pre, below) hands every tile its prefix: the decode kernel
//      then has no look-back at all and its tiles are independent (what was assumed is verified frame by frame),
//   4. integers are parsed from a right-aligned 12- or 20-byte field, four digits per 32-bit word (SWAR).
//
// Anything else — another tag, a table that is not eligible, a NULL in a NOT NULL column, text the lean parser does
// not take (a long run of leading zeros), a malformed frame, a transaction-state violation — sets
// DevResult.fused_fail bit 1: nothing of the batch is trusted and the host decodes it again with the generic
// kernels (fused.hip / cells.hip, then kernels.hip for the exact error cut). Errors end the stream in the reference
// (apply.rs:2475-2481), so that path is cold.
//
// Reference work replaced per row: LogicalReplicationMessage::parse (postgres-replication 0.6.7), handle_insert_message
// (apply.rs:2443-2491), convert_tuple_to_row (codec/event.rs:554-587), parse_cell_from_postgres_text for
// INT2/INT4/INT8/OID/BOOL (codec/text.rs:35-51,135-138, codec/bool.rs:11-19), payload accounting (codec/event.rs:261-297).
#include "lookback.hip.h"

namespace etlg {

// ---- aligned readers: `off` is a byte offset into the tile's window
struct WinLds {  // the staged tile; window byte 0 sits at a 16-byte aligned LDS address
  const ETLG_LDS_AS u8* base;
  DEV uint32_t w(uint32_t aoff) const { return ETLG_LDS_LD32(base + aoff); }
};
struct WinGlb {  // a tile that does not fit its LDS window reads the input in place (window byte 0 = input byte 0)
  const u8* base; uint32_t lim;
  DEV uint32_t w(uint32_t aoff) const {
    uint32_t v = 0;
    if (aoff + 4 <= lim) __builtin_memcpy(&v, base + aoff, 4);
    else for (uint32_t k = 0; aoff + k < lim; k++) v |= (uint32_t)base[aoff + k] << (8 * k);
    return v;
  }
};
template <class M> DEV uint32_t rd32(const M& m, uint32_t off) {
  const uint32_t a = off & ~3u, s = off & 3u;
  return __builtin_amdgcn_alignbyte(m.w(a + 4), m.w(a), s);
}
template <class M> DEV uint64_t rd64(const M& m, uint32_t off) {
  const uint32_t a = off & ~3u, s = off & 3u;
  const uint32_t w0 = m.w(a), w1 = m.w(a + 4), w2 = m.w(a + 8);
  return __builtin_amdgcn_alignbyte(w1, w0, s) | ((uint64_t)__builtin_amdgcn_alignbyte(w2, w1, s) << 32);
}
DEV uint64_t bswap64(uint64_t v) { return __builtin_bswap64(v); }

// Four ASCII digits, most significant in byte 0 (codec.hip.h digits4 without the flag plumbing): value, and a
// non-zero `bad` accumulator when a byte is not a digit.
DEV uint32_t dig4(uint32_t w, uint32_t& bad) {
  const uint32_t t = w - 0x30303030u;
  bad |= ((w + 0x46464646u) | t) & 0x80808080u;
  const uint32_t p = (t * 10u + (t >> 8)) & 0x00FF00FFu;
  return (p & 0xFFu) * 100u + (p >> 16);
}
// Word j of a field whose first k bytes are replaced by '0' (the text is right-aligned in the field). Branch-free.
DEV uint32_t pad_word(uint32_t w, uint32_t k, uint32_t j) {
  const int32_t sh = (int32_t)k - (int32_t)(4 * j);  // bytes of this word to replace, from byte 0
  const uint32_t part = 0xFFFFFFFFu << (8 * ((uint32_t)sh & 3u));
  const uint32_t keep = sh <= 0 ? 0xFFFFFFFFu : sh >= 4 ? 0u : part;
  return (w & keep) | (0x30303030u & ~keep);
}

// Rust iN::from_str / u32::from_str (codec/text.rs:40-51, 135-138) on the text [cs, cs + n) of the window, first
// character c0 already in hand. Straight-line code (every lane of the wave runs it, whatever its cell holds): takes
// what fits a 12-byte (32-bit classes) / 20-byte (int8) field and returns ok = 0 for everything else, INCLUDING every
// error — the generic kernels decide what a rejected text means. `cls` is wave-uniform.
template <class M>
DEV uint64_t plan_int(const M& m, uint32_t cs, uint32_t n, uint32_t c0, uint32_t cls, uint32_t& ok) {
  const uint32_t neg = c0 == '-' ? 1u : 0u;
  const uint32_t sg = (c0 == '-' || c0 == '+') ? 1u : 0u;
  const uint32_t nd = n - sg;  // digits
  uint32_t bad = 0;
  uint64_t mag;
  if (cls == ETLG_TC_I64) {
    const uint32_t nn = n < 20u ? n : 20u;   // the field never leaves the window, whatever the length says
    const uint32_t s0 = cs + nn - 20u, a = s0 & ~3u, s = s0 & 3u, k = 20u - (nd < 20u ? nd : 20u);
    const uint32_t w0 = m.w(a), w1 = m.w(a + 4), w2 = m.w(a + 8), w3 = m.w(a + 12), w4 = m.w(a + 16), w5 = m.w(a + 20);
    const uint32_t d0 = dig4(pad_word(__builtin_amdgcn_alignbyte(w1, w0, s), k, 0), bad);
    const uint32_t d1 = dig4(pad_word(__builtin_amdgcn_alignbyte(w2, w1, s), k, 1), bad);
    const uint32_t d2 = dig4(pad_word(__builtin_amdgcn_alignbyte(w3, w2, s), k, 2), bad);
    const uint32_t d3 = dig4(pad_word(__builtin_amdgcn_alignbyte(w4, w3, s), k, 3), bad);
    const uint32_t d4 = dig4(pad_word(__builtin_amdgcn_alignbyte(w5, w4, s), k, 4), bad);
    mag = (uint64_t)(d0 * 10000u + d1) * 1000000000000ull + (uint64_t)(d2 * 10000u + d3) * 10000ull + d4;  // nd <= 19: d0 <= 999, < 10^19
    const uint64_t lim = neg ? (1ull << 63) : (1ull << 63) - 1;
    ok = (n != 0u) & (nd != 0u) & (n <= 20u) & (nd <= 19u) & (bad == 0u) & (mag <= lim);
  } else {
    const uint32_t nn = n < 12u ? n : 12u;
    const uint32_t s0 = cs + nn - 12u, a = s0 & ~3u, s = s0 & 3u, k = 12u - (nd < 12u ? nd : 12u);
    const uint32_t w0 = m.w(a), w1 = m.w(a + 4), w2 = m.w(a + 8), w3 = m.w(a + 12);
    const uint32_t d0 = dig4(pad_word(__builtin_amdgcn_alignbyte(w1, w0, s), k, 0), bad);
    const uint32_t d1 = dig4(pad_word(__builtin_amdgcn_alignbyte(w2, w1, s), k, 1), bad);
    const uint32_t d2 = dig4(pad_word(__builtin_amdgcn_alignbyte(w3, w2, s), k, 2), bad);
    mag = (uint64_t)d0 * 100000000ull + (d1 * 10000u + d2);   // nd <= 11: d0 <= 999
    uint64_t lim;
    if (cls == ETLG_TC_U32) lim = 0xFFFFFFFFull;              // a '-' is rejected below: "-0" is an error for u32::from_str
    else if (cls == ETLG_TC_I16) lim = neg ? 0x8000ull : 0x7FFFull;
    else lim = neg ? 0x80000000ull : 0x7FFFFFFFull;
    ok = (n != 0u) & (nd != 0u) & (n <= 12u) & (nd <= 11u) & (bad == 0u) & (mag <= lim) & ((cls != ETLG_TC_U32) | (neg ^ 1u));
  }
  return neg ? 0 - mag : mag;
}

// Binary search of the (rel_id-sorted) plan tables with a wave-uniform key: scalar loads only.
DEV int plan_find(const PlanParams& q, uint32_t rel_u) {
  const ETLG_CONST_AS uint32_t* t = (const ETLG_CONST_AS uint32_t*)(uintptr_t)q.tabs;
  int lo = 0, hi = (int)q.n_tabs - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    const uint32_t v = t[mid * (int)(sizeof(PlanTab) / 4)];
    if (v == rel_u) return mid;
    if (v < rel_u) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

// ETLG_PLAN_DBG bit 5: per-phase shader-clock sums of one tile in 16 (lane 0), in DevResult.dbg_t[k]
// ETLG_PLAN_DBG bit 6: wall-clock (100 MHz, chip-wide) timeline of EVERY tile into the (otherwise unused) heap arena: 8 x u64 per tile
#define WSTAMP(k) do { if ((q.dbg & 64u) && threadIdx.x == 0) ((unsigned long long*)p.heap)[(size_t)blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
// ETLG_PLAN_DBG bit 5: per-phase shader-clock sums of one tile in 16 (lane 0), in DevResult.dbg_t[k]
#define PSTAMP(k) do { if ((q.dbg & 32u) && threadIdx.x == 0 && (blockIdx.x & 15u) == 3u) { const unsigned long long t_ = clock64(); atomicAdd(&p.res->dbg_t[k], t_ - tprev); tprev = t_; } } while (0)

// ---- the plan's look-back --------------------------------------------------------------------------------------------
// What a tile needs from the tiles before it: the sum of their fixed-arena dwords, the last Begin / Commit mark, and the LSN
// of the Begin that is open when the tile starts. One 16-byte descriptor per tile and per group of 64 tiles in the two-level
// decoupled look-back of lookback.hip.h: {status | mark : 30 | fixed dwords : 32, status | LSN : 62} — LSN of the range's last
// Begin when its last mark is one — read with ONE 16-byte load per lane and window; the fold carries the LSN along:
// f(older, newer) = {max mark, sum of dwords, newer has a mark ? newer's LSN : older's}. (Round 2 kept the LSN in a second
// array, read with a dependent load once the mark said which tile held the Begin: a third round trip per tile, ~5 us of the
// kernel.) Both halves carry the status, so a reader that catches a group descriptor between its AGG and INCL stores sees two
// different statuses and polls again; each half is written and read as one 8-byte unit at least.
// A round trip to another CU's words costs 1.5-2 us on this chip (such loads are served past the per-XCD L2s), so
// the order of work matters more than the instruction count: a tile PUBLISHES as soon as it has read its message heads,
// decodes its rows into LDS while its predecessors' words travel, and only then resolves. The last tile of a group of 64
// resolves right away instead: it is the one that folds the group descriptor the groups behind it are waiting for.
// The transaction state carried into the batch is NOT part of the fold (virtual group -1 holds the identity): a tile whose
// prefix holds no mark patches it in afterwards (plan_resolve) — so only the first tiles of a batch depend on the batch before
// it, and consecutive ASYNC batches can run side by side (DecParams.flags bit 4, host.cpp "two streams").
constexpr uint32_t kPreBeginLen = kBodyOff + 20u, kPreCommitLen = kBodyOff + 25u;   // CopyData + XLogData head + tag, then lsn:8 ts:8 xid:4 (Begin) / flags:1 lsn:8 end:8 ts:8 (Commit): 51 and 56 bytes
constexpr int kPreTilesPerWave = 16, kPreWaves = 16;   // the pre-pass: tiles a wave takes, waves of a workgroup
constexpr uint32_t kPreGroupLog = 8;                   // ... tiles per group = kPreTilesPerWave * kPreWaves = 1 << kPreGroupLog
constexpr uint32_t kPlanMaxPolls = 1u << 15;   // bounded spin (tens of milliseconds): a give-up sends the batch to the generic kernels

struct PlanPre2 { unsigned long long a0 = 0, l0 = 0, a1 = 0, l1 = 0; bool valid = false; };
struct PlanFold { uint32_t fx, mk, l0, l1; };   // fixed dwords, mark, LSN (low, high 30 bits)
DEV PlanFold fold_id() { return PlanFold{0u, 0u, 0u, 0u}; }
DEV PlanFold fold_f(const PlanFold& a, const PlanFold& b) {   // a older than b
  return PlanFold{a.fx + b.fx, a.mk > b.mk ? a.mk : b.mk, b.mk ? b.l0 : a.l0, b.mk ? b.l1 : a.l1};
}
DEV PlanFold fold_of(unsigned long long a, unsigned long long l) {
  return PlanFold{(uint32_t)a, (uint32_t)(a >> 32) & 0x3FFFFFFFu, (uint32_t)l, (uint32_t)(l >> 32) & 0x3FFFFFFFu};
}
DEV unsigned long long fold_agg(const PlanFold& v) { return ((unsigned long long)v.mk << 32) | v.fx; }
DEV unsigned long long fold_lsn(const PlanFold& v) { return ((unsigned long long)v.l1 << 32) | v.l0; }
DEV unsigned long long pair_state(unsigned long long a, unsigned long long l) { const unsigned long long st = a & ST_MASK; return st == (l & ST_MASK) ? st : 0ull; }
// inclusive scan over the wave, older lanes first (the DPP ladder of wave_scan_incl on four dwords)
DEV PlanFold fold_scan(PlanFold v) {
#define ETLG_P3_DPP(ctrl, rmask) { PlanFold o_; \
    o_.fx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.fx, ctrl, rmask, 0xF, false); o_.mk = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.mk, ctrl, rmask, 0xF, false); \
    o_.l0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l0, ctrl, rmask, 0xF, false); o_.l1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l1, ctrl, rmask, 0xF, false); \
    v = fold_f(o_, v); }
  ETLG_P3_DPP(0x111, 0xF) ETLG_P3_DPP(0x112, 0xF) ETLG_P3_DPP(0x114, 0xF) ETLG_P3_DPP(0x118, 0xF) ETLG_P3_DPP(0x142, 0xA) ETLG_P3_DPP(0x143, 0xC)
#undef ETLG_P3_DPP
  return v;
}
DEV PlanFold fold_lane63(const PlanFold& v) {
  return PlanFold{(uint32_t)__builtin_amdgcn_readlane((int)v.fx, 63), (uint32_t)__builtin_amdgcn_readlane((int)v.mk, 63),
                  (uint32_t)__builtin_amdgcn_readlane((int)v.l0, 63), (uint32_t)__builtin_amdgcn_readlane((int)v.l1, 63)};
}
DEV void plan_publish2(unsigned long long* d2, uint32_t tile, uint64_t agg, uint64_t lsn) {
  if (threadIdx.x == 0) ETLG_ST_PAIR(d2 + 2 * (size_t)tile, ST_AGG | agg, ST_AGG | lsn);
}
// the first words a resolve looks at (both windows), asked for ahead of time
DEV PlanPre2 plan_prefetch2(unsigned long long* d2, unsigned long long* g2, uint32_t tile) {
  const uint32_t lane = threadIdx.x, g = tile >> 6, j = tile & 63u;
  PlanPre2 r;
  r.valid = true;
  if (lane < j) ETLG_LD_PAIR(d2 + 2 * (size_t)((g << 6) + lane), r.a0, r.l0);
  const int64_t idx = (int64_t)g - 1 - 63 + (int64_t)lane;
  if (idx >= 0) ETLG_LD_PAIR(g2 + 2 * idx, r.a1, r.l1);
  return r;
}
// lookback_resolve (lookback.hip.h) on pairs, without a carry: the exclusive prefix of `tile` = {mark, fixed dwords} in `ex`, the open Begin's LSN in `ex_lsn`
DEV void plan_resolve2(unsigned long long* d2, unsigned long long* g2, uint32_t tile, uint64_t agg, uint64_t tile_lsn, uint32_t* fail,
                       const PlanPre2& pre, uint64_t& ex, uint64_t& ex_lsn) {
  const uint32_t lane = threadIdx.x, g = tile >> 6, j = tile & 63u;
  uint32_t polls = 0;
  ex = 0; ex_lsn = 0;
  unsigned long long a1p = pre.a1, l1p = pre.l1;
  if (!pre.valid) {
    const int64_t idx = (int64_t)g - 1 - 63 + (int64_t)lane;
    if (idx >= 0) ETLG_LD_PAIR(g2 + 2 * idx, a1p, l1p);
  }
  bool pre_valid = true;
  // ---- window 0: earlier tiles of this group
  unsigned long long a0 = pre.valid ? pre.a0 : 0ull, l0 = pre.valid ? pre.l0 : 0ull;
  bool have = lane >= j || pair_state(a0, l0) != 0;
  for (;;) {
    if (!have) {
      ETLG_LD_PAIR(d2 + 2 * (size_t)((g << 6) + lane), a0, l0);
      have = pair_state(a0, l0) != 0;
    }
    if (!__ballot(!have)) break;
    if (++polls > kPlanMaxPolls) { if (lane == 0) atomicOr(fail, 1u); return; }
    __builtin_amdgcn_s_sleep(2);
  }
  const PlanFold local = fold_lane63(fold_scan(lane < j ? fold_of(a0, l0) : fold_id()));   // lanes >= j hold the identity: lane 63 = fold of [0, j)
  const PlanFold own = fold_of(agg, tile_lsn);
  const PlanFold group_agg = fold_f(local, own);
  if (j == 63u && lane == 0) ETLG_ST_PAIR(g2 + 2 * (size_t)g, ST_AGG | fold_agg(group_agg), ST_AGG | fold_lsn(group_agg));
  // ---- window(s) 1: group descriptors before g, older groups in lower lanes
  PlanFold acc = fold_id();
  int64_t base = (int64_t)g - 1;
  for (;;) {
    const int64_t idx = base - 63 + (int64_t)lane;
    unsigned long long wa = ST_INCL, wl = ST_INCL;   // before group 0: the identity (the carried state is patched in by the caller)
    if (idx >= 0) {
      if (pre_valid) { wa = a1p; wl = l1p; }
      else ETLG_LD_PAIR(g2 + 2 * idx, wa, wl);
    }
    pre_valid = false;
    const unsigned long long st = pair_state(wa, wl);
    const unsigned long long m_incl = __ballot(st == ST_INCL);
    const unsigned long long m_empty = __ballot(st == 0);
    const int first_incl = m_incl ? 63 - __builtin_clzll(m_incl) : -1;
    const unsigned long long needed = first_incl <= 0 ? ~0ull : (~0ull << first_incl);
    if (m_empty & needed) {
      if (++polls > kPlanMaxPolls) { if (lane == 0) atomicOr(fail, 1u); return; }
      __builtin_amdgcn_s_sleep(2);
      continue;
    }
    const PlanFold wfold = fold_lane63(fold_scan((int)lane >= (first_incl < 0 ? 0 : first_incl) ? fold_of(wa, wl) : fold_id()));
    acc = fold_f(wfold, acc);
    if (first_incl >= 0) break;
    base -= 64;
  }
  if (j == 63u && lane == 0) { const PlanFold incl = fold_f(acc, group_agg); ETLG_ST_PAIR(g2 + 2 * (size_t)g, ST_INCL | fold_agg(incl), ST_INCL | fold_lsn(incl)); }
  const PlanFold r = fold_f(acc, local);
  ex = fold_agg(r); ex_lsn = fold_lsn(r);
}

// What a tile carries from its local phase (heads, publish, rows) to its finish (resolve, context, stores). A wave runs TWO tiles
// through one LDS window — local(A), local(B), finish(A), finish(B) — so A's state, its row image included, lives in registers.
template <int RD>
struct PlanLocal {
  uint32_t tbl, slot_id, pm, x_fx, fixed_dw, vbytes, bad;
  uint32_t tagf;           // tag | c_flags << 8 | isI << 16 | isB << 17 | isC << 18 | live << 19
  uint64_t wal_start, b_lsn;
  uint32_t* rimg;          // where the frame's body waits in LDS, or (the wave's first tile: the window is reused) ...
  uint32_t row[RD];        // ... the body itself (RD dwords: 6 or 8)
  uint64_t ord;            // written by plan_resolve (as is b_lsn: the commit_lsn column from then on)
  uint64_t agg, tile_lsn, ex, ex_lsn, tile_pre_lsn;   // wave-uniform from here on
  uint32_t tot_mark, tot_fx;
  bool early, done;
};

template <class M, int RD>
DEV void plan_local(const DecParams& p, const PlanParams& q, const M& m, u8* rows, bool stage_own, uint32_t a0, uint32_t tile, uint32_t nt, uint32_t my_o,
                    uint32_t span0, uint32_t span1, unsigned long long& tprev, PlanLocal<RD>& L, bool keep) {
  L.done = true;   // until the local phase has run to its end
  const uint32_t lane = threadIdx.x;
  const bool live = lane < nt;
  const uint32_t f = tile * 64u + lane;
  uint32_t* failp = &p.res->fused_fail;
  uint32_t bad = 0;  // non-zero: this lane saw something the plan does not cover (accumulated bitwise: straight-line code)

  // ---- (1) what the look-back needs, as early as possible: tag + table of every frame -> tile aggregate -> publish
  const uint32_t o0 = my_o;
  const uint32_t o1 = (uint32_t)__builtin_amdgcn_update_dpp((int)span1, (int)my_o, 0x130, 0xF, 0xF, false);  // wave_shl:1 — the next lane's offset
  const uint32_t flen = o1 - o0;
  const uint32_t sane = (live ? 1u : 0u) & (o1 > o0) & (o0 >= span0) & (o1 <= span1) & (flen >= 38u);
  bad |= (live ? 1u : 0u) & (sane ^ 1u);
  const uint32_t fr = sane ? o0 - a0 : 0u;  // window offset of the frame (a lane without a sane frame reads offset 0)
  const uint64_t h30 = rd64(m, fr + 30);    // tag | rel:4 | 'N' | ncols:2  (Insert)
  const uint32_t tag = (uint32_t)(h30 & 0xFF);
  const uint32_t rel = __builtin_bswap32((uint32_t)(h30 >> 8));
  const uint32_t ncols = ((uint32_t)(h30 >> 48) & 0xFFu) << 8 | (uint32_t)(h30 >> 56);
  bad |= sane & (uint32_t)(tag != 'I' && tag != 'U' && tag != 'D' && tag != 'B' && tag != 'C');
  // (round 6) an Update WITHOUT an old image — what pgoutput sends for a table under its default replica identity whenever the key did not
  // change — has the Insert's layout: rel | 'N' | tuple. Its new row is a full row (an unchanged-toast cell, 'u', is not the plan's:
  // it gives the batch up below like every cell that is neither 't' nor 'n'), its event differs in the kind byte and in the payload
  // counter it adds to. An Update that carries 'K' / 'O' fails the shape test ('N' behind the relation id) and goes the generic way.
  // (round 6, last session) a Delete BY KEY — rel | 'K' | tuple, what pgoutput sends under the default replica identity — decodes to the
  // table's key-layout row (normalize_key_tuple_to_row, codec/event.rs:795-923: a dense tuple of the identity columns, or a full-width
  // one whose other positions are skipped unread). 'O' (REPLICA IDENTITY FULL) and a table without identity columns go the generic way.
  const bool isD = sane && tag == 'D';
  const bool isI = sane && (tag == 'I' || tag == 'U'), isB = sane && tag == 'B', isC = sane && tag == 'C';
  // table of every Insert lane: one scalar lookup per distinct table of the wave (descriptors through the scalar cache)
  int ti = -1;
  uint32_t row_dw = 0, slot_id = 0, want_cols = 0, key_dw = 0, n_ident = 0;
  {
    unsigned long long todo = __ballot(isI || isD);
    while (todo) {
      const int leader = __builtin_ctzll(todo);
      const uint32_t rel_u = (uint32_t)__builtin_amdgcn_readlane((int)rel, leader);
      const int t_u = plan_find(q, rel_u);
      const bool mine = (isI || isD) && rel == rel_u;
      if (t_u >= 0) {
        const ETLG_CONST_AS uint32_t* tw = (const ETLG_CONST_AS uint32_t*)(uintptr_t)(q.tabs + t_u);
        const uint32_t s_u = tw[1], n_u = tw[2], r_u = tw[3], k_u = tw[5], i_u = tw[6];
        if (mine) { ti = t_u; slot_id = s_u; want_cols = n_u; row_dw = r_u; key_dw = k_u; n_ident = i_u; }
      }
      todo &= ~__ballot(mine);
    }
  }
  bad |= (uint32_t)(isI && (ti < 0 || ncols != want_cols));  // "Tuple data field count does not match schema" is the generic path's to report
  bad |= (uint32_t)(isD && (ti < 0 || key_dw == 0 || (ncols != n_ident && ncols != want_cols)));   // ("Replica-identity tuple shape does not match schema": likewise)
  uint64_t b_lsn = 0;  // Begin: final_lsn; Commit: commit_lsn
  if (isB) b_lsn = bswap64(rd64(m, fr + 31));
  if (isC) b_lsn = bswap64(rd64(m, fr + 32));
  const uint32_t fixed_dw = isI ? row_dw : isD ? key_dw : isB ? 2u : isC ? 4u : 0u;
  const uint32_t mark = isB ? (((f + 1) << 1) | 1u) : isC ? ((f + 1) << 1) : 0u;
  const uint32_t im = wave_scan_max(mark);
  const uint32_t pm = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)im, 0x138, 0xF, 0xF, false);  // wave_shr:1: exclusive running max
  const uint32_t ifx = wave_scan_add(fixed_dw);
  const uint32_t tot_mark = wave_last(im), tot_fx = wave_last(ifx);
  const uint32_t x_fx = ifx - fixed_dw;
  uint64_t tile_lsn = 0;  // LSN of the tile's last Begin, when its last mark is one
  if (tot_mark & 1u) {
    const int src = (int)((tot_mark >> 1) - 1u - tile * 64u);
    tile_lsn = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b_lsn >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b_lsn, src);
    bad |= (uint32_t)((tile_lsn >> 62) != 0);   // the LSN word keeps two bits for its own validity: such an LSN goes the generic way
    tile_lsn &= ~(3ull << 62);
  }
  PSTAMP(2);
  WSTAMP(2);
  if (q.dbg & 4u) { if (tot_mark + tot_fx == 0xFFFFFFF1u) atomicOr(failp, 2u); return; }  // profiling: stop after the message heads

  // ---- (2) publish, and while the predecessors' words travel: envelope checks, the rest of the heads, the rows into LDS
  //      (a row only needs the tile-local offset x_fx; where the tile's block goes in the arena is the look-back's answer)
  const uint64_t agg = ((uint64_t)tot_mark << 32) | tot_fx;
  const bool early = (tile & 63u) == 63u;   // the group's folder: the groups behind it wait for what it publishes next
  uint64_t ex = 0, ex_lsn = 0;
  // With the sidecar pre-pass (q.pre) the tile's prefix is a record in memory: nothing is published, nothing resolved. What the
  // pre-pass assumed about a frame it never read — a Begin is 51 bytes, a Commit 56, everything else a row of pre_row_dw dwords —
  // is checked here against the frame itself; a frame that breaks it sends the batch to the generic kernels like any other
  // shape the plan does not cover.
  // (a Delete by key was priced as one when it is shorter than any row frame can be, q.pre_key_below, or no longer than q.pre_key_max — its tag
  // was read then —, and as a row otherwise)
  if (q.pre) bad |= (uint32_t)((isB && flen != kPreBeginLen) || (isC && flen != kPreCommitLen) || (isI && ti >= 0 && (row_dw != q.pre_row_dw || flen < q.pre_key_below)) ||
                               (isD && ti >= 0 && (key_dw != q.pre_key_dw || (flen >= q.pre_key_below && flen > q.pre_key_max))));
  else plan_publish2(q.desc, tile, agg, tile_lsn);
  if (early && !q.pre) plan_resolve2(q.desc, q.desc + 2 * (size_t)q.ntiles, tile, agg, tile_lsn, failp, PlanPre2(), ex, ex_lsn);
  {
    const uint64_t h0 = rd64(m, fr);        // 'd' | len:4 | 'w' | 2 bytes of wal_start
    const uint32_t len = __builtin_bswap32((uint32_t)(h0 >> 8));
    const uint32_t env = ((uint32_t)(h0 & 0xFF) == 'd') & (len + 1u == flen) & ((uint32_t)((h0 >> 40) & 0xFF) == 'w');
    const uint32_t shape = (tag == 'I' || tag == 'U') ? (uint32_t)((uint32_t)((h30 >> 40) & 0xFF) == 'N')
                         : tag == 'D' ? (uint32_t)((uint32_t)((h30 >> 40) & 0xFF) == 'K')
                         : tag == 'B' ? (uint32_t)(flen >= kBodyOff + 20) : (uint32_t)(flen >= kBodyOff + 25);
    bad |= sane & ((env & shape) ^ 1u);
  }
  const uint64_t wal_start = bswap64(rd64(m, fr + 6));
  uint64_t b_ts = 0, c_end = 0;
  uint32_t b_xid = 0, c_flags = 0;
  if (isB) { b_ts = bswap64(rd64(m, fr + 39)); b_xid = __builtin_bswap32(rd32(m, fr + 47)); }
  if (isC) { c_flags = (uint32_t)(h30 >> 8) & 0xFFu; c_end = bswap64(rd64(m, fr + 40)); b_ts = bswap64(rd64(m, fr + 48)); }
  // Where a frame's body (its row; a Begin / Commit body) waits for the look-back. Up to 8 dwords fit the first 38 bytes of the
  // frame's OWN staged bytes — the message head, which is in registers by now — so narrow tables need no LDS beyond the window
  // (occupancy: ~20 waves per CU instead of 17). Wider rows take a separate region, at the tile-local arena offset.
  uint32_t* rimg = q.lds_bytes > q.rows_off ? (uint32_t*)(rows + q.rows_off) + x_fx : (uint32_t*)(rows + (stage_own ? ((fr + 3u) & ~3u) : lane * 32u));
  if (isB && !bad) { rimg[0] = (uint32_t)b_ts; rimg[1] = (uint32_t)(b_ts >> 32); }
  if (isC && !bad) { rimg[0] = (uint32_t)c_end; rimg[1] = (uint32_t)(c_end >> 32); rimg[2] = (uint32_t)b_ts; rimg[3] = (uint32_t)(b_ts >> 32); }
  uint32_t vbytes = 0;
  if (!(q.dbg & 8u)) {  // profiling: bit 3 skips the cell decode
    unsigned long long todo = __ballot(isI && !bad);
    while (todo) {  // one pass per distinct table of the wave (one, normally): column descriptors in SGPRs
      const int leader = __builtin_ctzll(todo);
      const int t_u = __builtin_amdgcn_readlane(ti, leader);
      const bool mine = isI && !bad && ti == t_u;
      const ETLG_CONST_AS uint32_t* tw = (const ETLG_CONST_AS uint32_t*)(uintptr_t)(q.tabs + t_u);
      const uint32_t n_u = tw[2], cb_u = tw[4];
      const ETLG_CONST_AS uint32_t* cw = (const ETLG_CONST_AS uint32_t*)(uintptr_t)(q.cols + cb_u);
      if (mine) {  // ONE divergent region; inside it the loop is wave-uniform and the cell code straight-line
        uint32_t cur = fr + 38;              // window offset of the cell being read
        const uint32_t e = fr + flen;
        uint32_t acc = 0, cbad = 0;
        uint32_t cd_next = cw[0];
        for (uint32_t i = 0; i < n_u; i++) {
          const uint32_t cd = cd_next;  // cls | nullable << 8 | off_full << 16
          cd_next = cw[i + 1 < n_u ? i + 1 : i];   // the next column's word travels while this cell is parsed
          const uint32_t cls = cd & 0xFFu;
          uint32_t* slot = rimg + (cd >> 18);
          const uint32_t rc = cur < e ? cur : fr;   // a cursor that ran off the frame is an error below; keep the reads inside the window
          const uint64_t ch = rd64(m, rc);          // kind | len:4 | first three characters
          const uint32_t kind = (uint32_t)(ch & 0xFF);
          const uint32_t len = __builtin_bswap32((uint32_t)(ch >> 8));
          const uint32_t c0 = (uint32_t)(ch >> 40) & 0xFFu;
          const uint32_t is_t = kind == 't', is_n = kind == 'n';
          const uint32_t fits = (cur < e) & (is_n | (is_t & (rc + 5 <= e) & (len <= e - rc - 5)));
          const uint32_t tl = (is_t & fits) ? len : 0u;   // text length the parsers may look at
          uint32_t okv;
          uint64_t v;
          if (cls == ETLG_TC_BOOL) {  // parse_bool, codec/bool.rs:11-19
            okv = (tl == 1u) & ((c0 == 't') | (c0 == 'f'));
            v = c0 == 't';
          } else {
            v = plan_int(m, rc + 5, tl, c0, cls, okv);
          }
          // NULL: convert_tuple_data_to_cell, codec/event.rs:945-961; anything but 't' / 'n' ('u' in a full row, 'b', garbage) is not the plan's
          const uint32_t good = fits & ((is_n & ((cd >> 8) & 1u)) | (is_t & okv));
          cbad |= good ^ 1u;
          const uint64_t val = is_t ? v : 0ull;
          slot[0] = (uint32_t)val;
          if (cls == ETLG_TC_I64) slot[1] = (uint32_t)(val >> 32);
          acc |= (is_n ? (uint32_t)ETLG_CELL_NULL : (uint32_t)ETLG_CELL_VALUE) << (2 * (i & 15));
          if ((i & 15) == 15 || i + 1 == n_u) { rimg[i >> 4] = acc; acc = 0; }
          vbytes += tl;
          cur = rc + (is_t ? 5u + tl : 1u);
        }
        bad |= cbad;
      }
      todo &= ~__ballot(mine);
    }
    // Deletes by key (rare: the loop body is skipped when the tile has none). The loop runs over the SCHEMA's columns, wave-uniform; a lane
    // reads a cell at a column when its tuple has one there — every column of a full-width tuple, the identity columns of a dense one.
    unsigned long long todo_d = __ballot(isD && !bad);
    while (todo_d) {
      const int leader = __builtin_ctzll(todo_d);
      const int t_u = __builtin_amdgcn_readlane(ti, leader);
      const bool mine = isD && !bad && ti == t_u;
      const ETLG_CONST_AS uint32_t* tw = (const ETLG_CONST_AS uint32_t*)(uintptr_t)(q.tabs + t_u);
      const uint32_t n_u = tw[2], cb_u = tw[4], ni_u = tw[6], kb_u = tw[7];
      const ETLG_CONST_AS uint32_t* cw = (const ETLG_CONST_AS uint32_t*)(uintptr_t)(q.cols + cb_u);
      const ETLG_CONST_AS uint32_t* kw = (const ETLG_CONST_AS uint32_t*)(uintptr_t)(q.cols + kb_u);
      if (mine) {
        const bool dense = ncols != n_u;   // (ncols == n_ident; a table whose every column is an identity column reads the same either way)
        uint32_t cur = fr + 38;
        const uint32_t e = fr + flen;
        uint32_t acc = 0, cbad = 0;
        for (uint32_t i = 0; i < n_u; i++) {
          const uint32_t cd = cw[i], kd = kw[i];
          const uint32_t ident = kd & 1u;
          if (!ident && dense) continue;            // a dense tuple has no cell for this column (wave-uniform per group of like tuples; lanes of the other kind go on)
          const uint32_t cls = cd & 0xFFu;
          const uint32_t rc = cur < e ? cur : fr;
          const uint64_t ch = rd64(m, rc);
          const uint32_t kind = (uint32_t)(ch & 0xFF);
          const uint32_t len = __builtin_bswap32((uint32_t)(ch >> 8));
          const uint32_t c0 = (uint32_t)(ch >> 40) & 0xFFu;
          const uint32_t is_t = kind == 't', is_n = kind == 'n';
          const uint32_t fits = (cur < e) & (is_n | (is_t & (rc + 5 <= e) & (len <= e - rc - 5)));
          const uint32_t tl = (is_t & fits) ? len : 0u;
          if (ident) {
            uint32_t okv;
            uint64_t v;
            if (cls == ETLG_TC_BOOL) { okv = (tl == 1u) & ((c0 == 't') | (c0 == 'f')); v = c0 == 't'; }
            else v = plan_int(m, rc + 5, tl, c0, cls, okv);
            const uint32_t good = fits & ((is_n & ((cd >> 8) & 1u)) | (is_t & okv));   // ('u' in a key tuple is "missing source value": the generic path's to report)
            cbad |= good ^ 1u;
            const uint64_t val = is_t ? v : 0ull;
            uint32_t* slot = rimg + (kd >> 24);
            const uint32_t k = (kd >> 8) & 0xFFFFu;
            slot[0] = (uint32_t)val;
            if (cls == ETLG_TC_I64) slot[1] = (uint32_t)(val >> 32);
            acc |= (is_n ? (uint32_t)ETLG_CELL_NULL : (uint32_t)ETLG_CELL_VALUE) << (2 * (k & 15));
            if ((k & 15) == 15 || k + 1 == ni_u) { rimg[k >> 4] = acc; acc = 0; }
          } else {
            cbad |= fits ^ 1u;   // a position the key row does not take: skipped unread, its bytes still count (calculate_tuple_bytes)
          }
          vbytes += tl;
          cur = rc + (is_t ? 5u + tl : 1u);
        }
        bad |= cbad;
      }
      todo_d &= ~__ballot(mine);
    }
  }
  PSTAMP(3);
  WSTAMP(3);
  // (storing the header columns that do not depend on the look-back here, early, was tried: the descriptor loads of the finish
  // phase then queue behind those stores — memory operations of a wave return in order — and the kernel got 4 us slower)
  L.tbl = isB ? b_xid : (isI || isD) ? rel : 0u; L.slot_id = (isI || isD) ? slot_id : 0u; L.pm = pm; L.x_fx = x_fx; L.fixed_dw = fixed_dw; L.vbytes = vbytes; L.bad = bad;
  if (isD) c_flags = ETLG_OLD_KEY;   // the event's flags column: which old row a Delete carries (write_frame, codec.hip.h)
  L.tagf = tag | (c_flags << 8) | (((isI || isD) ? 1u : 0u) << 16) | ((isB ? 1u : 0u) << 17) | ((isC ? 1u : 0u) << 18) | ((live ? 1u : 0u) << 19);
  L.wal_start = wal_start; L.b_lsn = b_lsn; L.rimg = rimg;
  if (keep) {
#pragma unroll
    for (int d = 0; d < RD; d++) L.row[d] = rimg[d];   // inside the frame's own 38-byte head (or the lane's 32 bytes): always readable
  }
  L.agg = agg; L.tile_lsn = tile_lsn; L.ex = ex; L.ex_lsn = ex_lsn; L.tot_mark = tot_mark; L.tot_fx = tot_fx; L.early = early; L.done = false;
}

// The transaction state a batch starts from when the batch before it may still be running (DecParams.flags bit 4, set by the
// host for batches it puts on the second stream): wait for that batch's last tile to have written its totals, then read them.
// Wave-uniform; once per wave. No poison check here: when a batch of a chain fails, the host decodes its successors again.
DEV void plan_late_carry(DecParams& p, uint32_t* failp) {
  if (!(p.flags & 16u) || !p.carry) return;
  for (uint32_t polls = 0;; polls++) {
    if (__hip_atomic_load(&p.carry->carry_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) break;
    if (polls > kPlanMaxPolls) { if (threadIdx.x == 0) atomicOr(failp, 1u); break; }
    __builtin_amdgcn_s_sleep(8);
  }
  p.in_txn = __hip_atomic_load(&p.carry->out_in_txn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  p.final_lsn = __hip_atomic_load(&p.carry->out_final_lsn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  p.next_ord = __hip_atomic_load(&p.carry->out_next_ord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  p.flags &= ~16u;
}

// Finish, part 1: everything that LOADS — the look-back's answer, with it the open Begin's LSN — and the transaction context. A wave
// runs this for both of its tiles before it stores anything: memory operations of a wave return in order, so a descriptor load issued
// behind a tile's row / header stores waits for those stores to be acknowledged first.
// The two words the sidecar pre-pass left for a tile — its prefix inside its group of 256 (kPreGroupLog), the group's prefix — through scalar loads
// (one address per wave; written by the kernels before this one on the stream), requested before the tile's bytes are.
struct PlanPreWords { unsigned long long g0 = 0, g1 = 0, t0 = 0, t1 = 0; };
DEV PlanPreWords plan_pre_load(const PlanParams& q, uint32_t tile) {
  PlanPreWords w;
  if (q.pre && tile < q.ntiles) {
    const ETLG_CONST_AS unsigned long long* dt = (const ETLG_CONST_AS unsigned long long*)(uintptr_t)(q.pre + 2 * (size_t)tile);
    const ETLG_CONST_AS unsigned long long* dg = (const ETLG_CONST_AS unsigned long long*)(uintptr_t)(q.pre + 2 * ((size_t)q.ntiles + (tile >> kPreGroupLog)));
    w.t0 = dt[0]; w.t1 = dt[1]; w.g0 = dg[0]; w.g1 = dg[1];
  }
  return w;
}

template <int RD>
DEV void plan_resolve(DecParams& p, const PlanParams& q, uint32_t tile, PlanLocal<RD>& L, unsigned long long& tprev, const PlanPre2& pre = PlanPre2(), const PlanPreWords& pw = PlanPreWords()) {
  if (L.done) return;
  const uint32_t lane = threadIdx.x;
  const uint32_t f = tile * 64u + lane;
  uint32_t* failp = &p.res->fused_fail;
  uint32_t bad = L.bad;
  const bool isI = (L.tagf >> 16) & 1u, isB = (L.tagf >> 17) & 1u, isC = (L.tagf >> 18) & 1u;
  const uint32_t pm = L.pm;
  const uint64_t b_lsn = L.b_lsn;
  uint64_t ex = L.ex, ex_lsn = L.ex_lsn;

  // ---- (3) the look-back's answer
  if (q.pre) {   // the pre-pass left the tile's prefix inside its group and the group's prefix: asked for at the kernel's start (plan_pre_load)
    const PlanFold r = fold_f(fold_of(pw.g0, pw.g1), fold_of(pw.t0, pw.t1));
    ex = fold_agg(r); ex_lsn = fold_lsn(r);
  }
  else if (!L.early) plan_resolve2(q.desc, q.desc + 2 * (size_t)q.ntiles, tile, L.agg, L.tile_lsn, failp, pre, ex, ex_lsn);
  PSTAMP(7);
  uint32_t pre_mark = (uint32_t)(ex >> 32);
  if (pre_mark == 0) {   // no Begin / Commit before this tile in the batch: the state the batch started from decides (virtual Begin before frame 0)
    plan_late_carry(p, failp);
    if (p.in_txn) { pre_mark = 1u; ex |= 1ull << 32; }
  }
  // LSN of the Begin that is open when this tile starts: carried in, or (it lives in an earlier tile) it came with the fold
  const uint64_t pre_lsn = (pre_mark & 1u) && (pre_mark >> 1) != 0 ? ex_lsn : p.final_lsn;
  PSTAMP(4);
  WSTAMP(4);

  // ---- transaction context of every frame (A2: apply.rs:2279-2617, ordinals :942-963)
  const uint32_t last = pre_mark > pm ? pre_mark : pm;  // last Begin / Commit strictly before this frame
  const bool in_txn = (last & 1u) != 0;
  const uint32_t fb1 = last >> 1;                       // frame index of that Begin + 1; 0 = carried in from an earlier batch
  uint64_t final_lsn = 0;
  {
    // a Begin inside this tile: its lane holds the LSN; one from an earlier tile came with the look-back (wave-uniform)
    const bool here = in_txn && fb1 != 0 && fb1 - 1 >= tile * 64u;
    const uint32_t src = here ? fb1 - 1 - tile * 64u : 0u;
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)b_lsn, (int)src, 64), hi = (uint32_t)__shfl((int)(uint32_t)(b_lsn >> 32), (int)src, 64);
    if (in_txn) final_lsn = here ? (((uint64_t)hi << 32) | lo) : pre_lsn;
  }
  uint64_t ord = 0;
  if (!isB && in_txn) ord = fb1 == 0 ? p.next_ord + f : (uint64_t)(f - (fb1 - 1));
  bad |= (uint32_t)((isI || isC) && !in_txn);               // "Invalid transaction state"
  bad |= (uint32_t)(isC && in_txn && b_lsn != final_lsn);   // "Invalid commit LSN"
  L.bad = bad; L.ex = ex; L.tile_pre_lsn = pre_lsn;
  L.b_lsn = isI ? final_lsn : b_lsn;   // from here on: the event's commit_lsn column
  L.ord = ord;
}

// Finish, part 2: everything that STORES.
template <bool REGS, int RD>
DEV void plan_store(DecParams& p, const PlanParams& q, uint32_t tile, const PlanLocal<RD>& L, unsigned long long& tprev) {
  if (L.done) return;
  const uint32_t lane = threadIdx.x;
  const uint32_t f = tile * 64u + lane;
  uint32_t* failp = &p.res->fused_fail;
  const uint32_t bad = L.bad;
  const uint32_t tag = L.tagf & 0xFFu, c_flags = (L.tagf >> 8) & 0xFFu;
  const bool live = (L.tagf >> 19) & 1u;
  const uint32_t x_fx = L.x_fx, fixed_dw = L.fixed_dw, vbytes = L.vbytes, tot_mark = L.tot_mark, tot_fx = L.tot_fx;
  const uint64_t tile_lsn = L.tile_lsn, pre_lsn = L.tile_pre_lsn;
  const uint32_t pre_mark = (uint32_t)(L.ex >> 32);
  const uint64_t pre_fx = (uint64_t)(uint32_t)L.ex << 2;  // bytes

  // ---- totals and the carried transaction state (last tile)
  if (tile == q.ntiles - 1 && lane == 0) {
    DevResult* r = p.res;
    r->n_events = p.nframes; r->fixed_bytes = pre_fx + ((uint64_t)tot_fx << 2); r->heap_bytes = 0; r->n_frames = p.nframes;
    const uint32_t lm = pre_mark > tot_mark ? pre_mark : tot_mark;
    const bool it = (lm & 1u) != 0;
    r->out_in_txn = it;
    uint64_t fl = 0, no = 0;
    if (it) {
      const uint32_t lb1 = lm >> 1;
      fl = tot_mark ? tile_lsn : pre_lsn;   // marks grow with the frame index: a mark of this tile is the last one
      no = lb1 == 0 ? p.next_ord + p.nframes : (uint64_t)(p.nframes - (lb1 - 1));
    }
    r->out_final_lsn = fl; r->out_next_ord = no;
    __hip_atomic_store(&r->carry_ready, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // the batch behind this one may be waiting for exactly these three words (plan_late_carry)
  }

  // ---- (4) rows and Begin / Commit bodies: LDS -> their final place in the fixed arena
  const uint64_t fx_off = pre_fx + ((uint64_t)x_fx << 2);
  const uint32_t any_bad = __any(bad != 0u) ? 1u : 0u;
  const bool cap_ok = pre_fx + ((uint64_t)tot_fx << 2) <= p.fixed_cap;
  if (!any_bad && cap_ok && !(q.dbg & 8u)) {
    uint32_t* dst = (uint32_t*)(p.fixed + fx_off);
    if (REGS) {
#pragma unroll
      for (uint32_t d = 0; d < (uint32_t)RD; d++) if (d < fixed_dw) dst[d] = L.row[d];
    } else {
      const uint32_t* rimg = L.rimg;
      for (uint32_t d = 0; d < fixed_dw; d++) dst[d] = rimg[d];
    }
  }
  PSTAMP(5);
  // ---- event headers (A13/A15)
  if (live && !any_bad && cap_ok && !(q.dbg & 16u)) {  // profiling: bit 4 skips the event header stores
    p.ev_kind[f] = (u8)tag;
    p.ev_flags[f] = (u8)c_flags;
    p.ev_table[f] = L.tbl;
    p.ev_slot[f] = L.slot_id;
    p.ev_start[f] = L.wal_start;
    p.ev_commit[f] = L.b_lsn;
    p.ev_ord[f] = L.ord;
    p.ev_body[f] = fx_off;
  }
  PSTAMP(6);
  WSTAMP(5);
  // payload bytes of the tile's inserts / updates (A3), one atomic per wave into a shard
  const bool is_upd = (L.tagf & 0xFFu) == 'U', is_del = (L.tagf & 0xFFu) == 'D';
  const uint32_t pay = wave_last(wave_scan_add((is_upd || is_del) ? 0u : vbytes));
  if (lane == 0 && pay) atomicAdd(&p.res->pay_shard[tile & 31][0], (unsigned long long)pay);
  if (__ballot(is_upd && vbytes != 0)) {
    const uint32_t payu = wave_last(wave_scan_add(is_upd ? vbytes : 0u));
    if (lane == 0 && payu) atomicAdd(&p.res->pay_shard[tile & 31][1], (unsigned long long)payu);
  }
  if (__ballot(is_del && vbytes != 0)) {
    const uint32_t payd = wave_last(wave_scan_add(is_del ? vbytes : 0u));
    if (lane == 0 && payd) atomicAdd(&p.res->pay_shard[tile & 31][2], (unsigned long long)payd);
  }
  if ((any_bad || !cap_ok) && lane == 0) atomicOr(failp, 2u);
}

// A tile's frames: their count, the byte span (two scalar loads) and this lane's offset
struct TileSpan { uint32_t nt, span0, span1, my_o; };
DEV TileSpan plan_span(const DecParams& p, uint32_t tile) {
  const uint32_t lane = threadIdx.x;
  TileSpan t;
  const uint32_t f0 = tile * 64u;
  t.nt = p.nframes - f0 < 64u ? p.nframes - f0 : 64u;
  const ETLG_CONST_AS uint32_t* offs_c = (const ETLG_CONST_AS uint32_t*)(uintptr_t)p.offs;
  t.span0 = offs_c[f0]; t.span1 = offs_c[f0 + t.nt];
  t.my_o = lane < t.nt ? p.offs[f0 + lane] : t.span1;
  return t;
}

// Stages one tile (LDS-DMA) and runs its local phase.
template <int RD>
DEV void plan_stage_local(DecParams& p, const PlanParams& q, u8* smem, uint32_t tile, const TileSpan& ts, unsigned long long& tprev, PlanLocal<RD>& L, bool keep) {
  const uint32_t lane = threadIdx.x;
  L.done = true;
  const uint32_t nt = ts.nt, span0 = ts.span0, span1 = ts.span1, my_o = ts.my_o;
  PSTAMP(0);
  const uint32_t a0 = span0 & ~15u;
  const bool staged = span1 > span0 && span1 <= p.in_len && (uint64_t)(span1 - a0) + 64 <= q.rows_off && !(q.dbg & 1u);
  if (staged) {
    // LDS-DMA: piece k of the window = 1 KiB, lane l moves bytes [a0 + 1024 k + 16 l, +16) to LDS offset 1024 k + 16 l.
    // A 16-byte piece that would cross the end of the input is moved bytewise by the first lanes instead.
    const uint32_t npieces = (span1 - a0 + 1023u) >> 10;
    for (uint32_t k = 0; k < npieces; k++) {
      const uint32_t c = a0 + (k << 10) + (lane << 4);
      if (c < span1 && (uint64_t)c + 16 <= p.in_len) ETLG_GLDS16(p.in + c, smem + (k << 10));
    }
    const uint32_t tail = a0 + (((uint32_t)p.in_len - a0) & ~15u);  // the piece holding the last input byte, if partial
    if (tail < span1 && (uint64_t)tail + 16 > p.in_len && tail + lane < p.in_len && lane < 16) smem[tail - a0 + lane] = p.in[tail + lane];
    ETLG_VMEM_WAIT();
    PSTAMP(1);
    WSTAMP(1);
    if (q.dbg & 2u) { if (smem[lane * 97u] == 0xEE && smem[lane * 13u + 5u] == 0xEF && my_o == 0xFFFFFFF1u) atomicOr(&p.res->fused_fail, 2u); return; }  // profiling: staging only
    const WinLds m{(const ETLG_LDS_AS u8*)smem};
    plan_local(p, q, m, smem, true, a0, tile, nt, my_o, span0, span1, tprev, L, keep);
  } else {
    const WinGlb m{p.in, (uint32_t)p.in_len};
    plan_local(p, q, m, smem, false, 0u, tile, nt, my_o, span0, span1, tprev, L, keep);
  }
}

// A batch whose record-boundary scan was still running when this launch was enqueued (DecParams.nframes_dev): the frame count is read from
// the device now; p.nframes / q.ntiles were the bound the grid was sized by. Returns false when there is nothing to do here — the scan did
// not hold, the batch is empty or larger than the bound (DevResult.fused_fail bit 5: the host collects the scan and decodes the batch again
// with the count in hand) — after letting a batch that may be polling for this one's carried state go on (it is decoded again as well).
DEV bool plan_frames_from_device(DecParams& p, PlanParams& q) {
  if (!p.nframes_dev) return true;
  const ETLG_CONST_AS uint32_t* r = (const ETLG_CONST_AS uint32_t*)(uintptr_t)p.nframes_dev;
  const uint32_t nf = r[0], flags = r[1] | r[2];   // ([2]: tiles of the scan whose guess did not hold — the boundaries are not the chain's)
  if (flags != 0 || nf == 0 || nf > p.nframes) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      atomicOr(&p.res->fused_fail, 32u);
      __hip_atomic_store(&p.res->carry_ready, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    return false;
  }
  p.nframes = nf;
  q.ntiles = (nf + 63u) >> 6;
  return true;
}

// TWO: a wave takes two consecutive tiles through ONE LDS window — local(A), local(B), finish(A), finish(B): half as many waves
// as tiles (a 64 MiB cfg2 batch is 4 645 waves: one round of the chip's ~5 100 slots instead of 1.8), and A's look-back is resolved
// a whole tile's work after it was published. Only for rows of up to 8 dwords (A's image waits in registers).
template <bool TWO, int RD>
DEV void plan_kernel(DecParams& p, const PlanParams& q_in, u8* smem) {
  PlanParams q = q_in;
  const uint32_t lane = threadIdx.x;
  unsigned long long tprev = (q.dbg & 32u) ? clock64() : 0ull;
  const uint32_t tile = blockIdx.x;   // (the timeline's slot)
  WSTAMP(0);
  if (q.clear_words) {  // descriptors are double buffered: this launch clears the buffer the next batch will use
    const uint32_t per = (q.clear_words + gridDim.x - 1) / gridDim.x;
    for (uint32_t i = lane; i < per; i += 64) { const uint32_t w = blockIdx.x * per + i; if (w < q.clear_words) q.d_clear[w] = 0; }
  }
  // ASYNC chain: the state the batch before this one left on the device — read now (that batch has finished: same stream), or, for
  // a batch the host put on the second stream, by the few tiles that need it, when they need it (plan_late_carry)
  if (!(p.flags & 16u) && !load_carry(p)) return;
  if (!plan_frames_from_device(p, q)) return;
  const uint32_t t0 = TWO ? 2u * blockIdx.x : blockIdx.x;
  if (t0 >= q.ntiles) return;   // (a grid sized by a bound)
  const TileSpan sa = plan_span(p, t0);
  const PlanPreWords pwa = plan_pre_load(q, t0), pwb = TWO ? plan_pre_load(q, t0 + 1u) : PlanPreWords();
  TileSpan sb = sa;
  if (TWO && t0 + 1u < q.ntiles) sb = plan_span(p, t0 + 1u);   // B's offsets travel while A is worked on
  PlanLocal<RD> A;
  plan_stage_local(p, q, smem, t0, sa, tprev, A, TWO);
  if (TWO) {
    const uint32_t t1 = t0 + 1u;
    PlanLocal<RD> B;
    B.done = true;
    if (t1 < q.ntiles) {
      __threadfence_block();
      __syncthreads();   // every read of A's window has returned before B's bytes land in it
      plan_stage_local(p, q, smem, t1, sb, tprev, B, false);
    }
    // both tiles' first look-back words are requested before the first wait: one round trip instead of two
    PlanPre2 preA, preB;
    if (!A.done && !A.early && !q.pre) preA = plan_prefetch2(q.desc, q.desc + 2 * (size_t)q.ntiles, t0);
    if (!B.done && !B.early && !q.pre) preB = plan_prefetch2(q.desc, q.desc + 2 * (size_t)q.ntiles, t1);
    plan_resolve(p, q, t0, A, tprev, preA, pwa);
    plan_resolve(p, q, t1, B, tprev, preB, pwb);
    plan_store<true>(p, q, t0, A, tprev);
    plan_store<false>(p, q, t1, B, tprev);
  } else {
    plan_resolve(p, q, t0, A, tprev, PlanPre2(), pwa);
    plan_store<false>(p, q, t0, A, tprev);
  }
}

// ---- the sidecar pre-pass ----------------------------------------------------------------------------------------------------
// The look-back is what makes a plan tile wait: two or three dependent round trips past the L2s (1.5-2 us each) in the middle of
// every tile, ~10 of the kernel's 51 us on cfg2, and the reason the batch runs as one lock-step round (profiles/r04ad: the same
// kernel with its prefixes given is 37 us, one tile per wave). But what a plan tile needs from its predecessors — fixed-arena dwords,
// the last Begin / Commit, the open Begin's LSN — is a function of WHICH frames are Begins and Commits, and when every planned table
// has the same row size that can be read off the offsets sidecar: pgoutput's Begin is kPreBeginLen bytes on the wire, its Commit
// kPreCommitLen, and a frame of another length is priced as a row. So a small kernel runs the look-back AHEAD of the decode — one wave
// per tile over the sidecar (3.5 % of the input's bytes; the tag byte is read only for the frames of those two lengths, the LSN
// only for Begins), same descriptors, same fold — and leaves every tile's exclusive prefix in memory; the decode kernel picks its
// record up with the tile's offsets and never talks to another tile. It also runs beside the decode of the batch BEFORE it (the two
// decode streams): its cost leaves the chain's critical path. The decode kernel verifies what the pre-pass assumed frame by frame
// (plan_local): a stream that breaks it — a Begin with trailing bytes, a row of another table size, anything that is not B / C / I —
// gives the batch up to the generic kernels exactly like today.
// k_plan_pre: one workgroup of sixteen waves per GROUP of 256 tiles, a wave takes 16 consecutive tiles (lane = frame, tile after tile; the
// offsets of all 16 are requested before the first is looked at). Leaves desc[tile] = the tile's exclusive prefix INSIDE its group and,
// behind them, gdesc[group] = the exclusive prefix of the group: every workgroup stores its group's
// aggregate (16-byte word, this launch's status tag in both halves) and takes a ticket; the LAST one to arrive turns the aggregates
// into exclusive prefixes in place — one wave, 64 groups per trip, the fold carried from trip to trip, a word that has not landed yet
// polled like a look-back word (no fences: status and payload travel in one store). A decode tile folds the two words it owns: two
// scalar loads, no other tile.
__global__ __launch_bounds__(kPreWaves * 64) void k_plan_pre(DecParams p, PlanParams q) {
  __shared__ unsigned long long s_wagg[kPreWaves][2];
  if (!plan_frames_from_device(p, q)) return;   // (the decode kernel behind this one reports it)
  if (blockIdx.x >= ((q.ntiles + (1u << kPreGroupLog) - 1u) >> kPreGroupLog)) return;
  const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), g = blockIdx.x;
  const uint32_t t0 = (g << kPreGroupLog) + wave * (uint32_t)kPreTilesPerWave;
  // offsets: lane l of iteration i owns frame (t0 + i) * 64 + l; its end is the next lane's start (the next iteration's lane 0 for lane 63)
  uint32_t o[kPreTilesPerWave + 1];
#pragma unroll
  for (int i = 0; i <= kPreTilesPerWave; i++) {
    const uint64_t f = (uint64_t)(t0 + (uint32_t)i) * 64u + lane;
    o[i] = p.offs[f < p.nframes ? f : p.nframes];
  }
  // the frames of a Begin's or a Commit's length: their tag byte and the eight bytes behind it, all 16 tiles' requests in flight at once
  // (one round trip; a wave meets a handful of such frames, and waiting for each tile's on its own was most of this kernel)
  uint32_t cand_m = 0;          // bit i: this lane's frame of tile i is such a frame
  uint32_t short_m = 0;         // bit i: ... is shorter than a row frame can be
  uint32_t tagv[kPreTilesPerWave];
  uint64_t lsnv[kPreTilesPerWave];
#pragma unroll
  for (int i = 0; i < kPreTilesPerWave; i++) {
    const uint64_t f = (uint64_t)(t0 + (uint32_t)i) * 64u + lane;
    const uint32_t o0 = o[i];
    const uint32_t nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)o[i + 1]);   // lane 0 of the next tile
    const uint32_t o1 = (uint32_t)__builtin_amdgcn_update_dpp((int)nxt, (int)o0, 0x130, 0xF, 0xF, false);  // wave_shl:1 — the next lane's offset
    const uint32_t flen = o1 - o0;
    const bool cand = f < p.nframes && o1 > o0 && o1 <= p.in_len && (flen == kPreBeginLen || flen == kPreCommitLen || (flen >= q.pre_key_below && flen <= q.pre_key_max));
    tagv[i] = 0; lsnv[i] = 0;
    if (cand) { cand_m |= 1u << i; tagv[i] = p.in[o0 + 30] | (flen << 8); lsnv[i] = ld_be64(p.in + o0 + 31); }   // (a Commit's bytes there are its flags and seven of its LSN: unused)
    if (flen < q.pre_key_below) short_m |= 1u << i;   // shorter than any row frame: a Delete by key, unless it is a Begin / Commit (the decode kernel checks each)
  }
  PlanFold run = fold_id();     // fold of this wave's tiles so far (wave-uniform)
  PlanFold keep = fold_id();    // lane i: the exclusive prefix of tile t0 + i inside the wave
#pragma unroll
  for (int i = 0; i < kPreTilesPerWave; i++) {
    const uint32_t tile = t0 + (uint32_t)i;
    const uint64_t f = (uint64_t)tile * 64u + lane;
    const bool live = f < p.nframes;
    const bool cand = (cand_m >> i) & 1u;
    const uint32_t tag = tagv[i] & 0xFFu, flen = tagv[i] >> 8;
    const uint64_t lsn = lsnv[i];
    const bool isB = cand && flen == kPreBeginLen && tag == 'B', isC = cand && flen == kPreCommitLen && tag == 'C';
    const bool isDel = ((short_m >> i) & 1u) || (cand && tag == 'D' && flen <= q.pre_key_max);   // (pre_key_max 0: never)
    const uint32_t fixed_dw = !live ? 0u : isB ? 2u : isC ? 4u : isDel ? q.pre_key_dw : q.pre_row_dw;
    const uint32_t mark = isB ? ((((uint32_t)f + 1u) << 1) | 1u) : isC ? (((uint32_t)f + 1u) << 1) : 0u;
    const uint32_t tot_mark = wave_last(wave_scan_max(mark)), tot_fx = wave_last(wave_scan_add(fixed_dw));
    uint64_t tile_lsn = 0;
    if (tot_mark & 1u) {
      const int src = (int)((tot_mark >> 1) - 1u - tile * 64u);
      tile_lsn = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(lsn >> 32), src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)lsn, src);
      tile_lsn &= ~(3ull << 62);   // (an LSN with those bits set fails the decode kernel's own check)
    }
    if (lane == (uint32_t)i) keep = run;
    run = fold_f(run, PlanFold{tot_fx, tot_mark, (uint32_t)tile_lsn, (uint32_t)(tile_lsn >> 32)});
  }
  if (lane == 0) { s_wagg[wave][0] = fold_agg(run); s_wagg[wave][1] = fold_lsn(run); }
  __syncthreads();
  // the waves of this group in front of this one: every wave scans the sixteen aggregates (lane = wave) and reads its own prefix off
  const PlanFold winc = fold_scan(lane < (uint32_t)kPreWaves ? fold_of(s_wagg[lane][0], s_wagg[lane][1]) : fold_id());
  PlanFold before = fold_id();
  if (wave) {
    const int src = (int)wave - 1;
    before = PlanFold{(uint32_t)__builtin_amdgcn_readlane((int)winc.fx, src), (uint32_t)__builtin_amdgcn_readlane((int)winc.mk, src),
                      (uint32_t)__builtin_amdgcn_readlane((int)winc.l0, src), (uint32_t)__builtin_amdgcn_readlane((int)winc.l1, src)};
  }
  if (lane < (uint32_t)kPreTilesPerWave && t0 + lane < q.ntiles) {
    const PlanFold r = fold_f(before, keep);
    unsigned long long* d = q.pre_out + 2 * (size_t)(t0 + lane);
    d[0] = fold_agg(r); d[1] = fold_lsn(r);
  }
  if (wave != (uint32_t)kPreWaves - 1u) return;
  // ---- the group's aggregate, a ticket, and (the last group to arrive) the scan over all of them
  const uint32_t ng = (q.ntiles + (1u << kPreGroupLog) - 1u) >> kPreGroupLog;
  unsigned long long* gd = q.pre_out + 2 * (size_t)q.ntiles;
  const unsigned long long tagbits = (unsigned long long)q.pre_tag << 62;
  uint32_t ticket = 0;
  if (lane == 0) {
    const PlanFold all = fold_f(before, run);
    ETLG_ST_PAIR(gd + 2 * (size_t)g, tagbits | fold_agg(all), tagbits | fold_lsn(all));
    ticket = atomicAdd(q.pre_ticket, 1u);
  }
  ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
  if (ticket != ng - 1u) return;
  PlanFold carry = fold_id();
  unsigned long long na = 0, nl = 0;   // the next trip's words, requested a trip ahead
  if (lane < ng) ETLG_LD_PAIR(gd + 2 * (size_t)lane, na, nl);
  for (uint32_t base = 0; base < ng; base += 64u) {
    const uint32_t gi = base + lane;
    unsigned long long a = gi < ng ? na : tagbits, l = gi < ng ? nl : tagbits;   // lanes past the last group: the identity, present
    if (gi + 64u < ng) ETLG_LD_PAIR(gd + 2 * (size_t)(gi + 64u), na, nl);
    bool have = pair_state(a, l) == tagbits;
    bool gave_up = false;
    for (uint32_t polls = 0;; polls++) {
      if (!have) { ETLG_LD_PAIR(gd + 2 * (size_t)gi, a, l); have = pair_state(a, l) == tagbits; }
      if (!__ballot(!have)) break;
      if (polls > kPlanMaxPolls) { if (lane == 0) atomicOr(&p.res->fused_fail, 1u); gave_up = true; break; }   // (the decode kernel's result is discarded with it)
      __builtin_amdgcn_s_sleep(2);
    }
    const PlanFold inc = fold_scan(gi < ng ? fold_of(a, l) : fold_id());
    PlanFold exc;   // previous lane's inclusive value (wave_shr:1), the identity into lane 0
    exc.fx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc.fx, 0x138, 0xF, 0xF, false); exc.mk = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc.mk, 0x138, 0xF, 0xF, false);
    exc.l0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc.l0, 0x138, 0xF, 0xF, false); exc.l1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc.l1, 0x138, 0xF, 0xF, false);
    const PlanFold r = fold_f(carry, exc);
    // (status 0: never this or a later launch's tag. After a give-up nothing is rewritten: the words keep this launch's tag, which no
    // later use of the buffer before the tag's next turn takes for its own, and the batch is decoded again anyway)
    if (gi < ng && !gave_up) { gd[2 * (size_t)gi] = fold_agg(r); gd[2 * (size_t)gi + 1] = fold_lsn(r); }
    carry = fold_f(carry, fold_lane63(inc));
  }
  if (lane == 0) __hip_atomic_store(q.pre_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // as the next user of the buffer expects it
}

__global__ __launch_bounds__(64, ETLG_PLAN_MINWAVES) void k_plan(DecParams p, PlanParams q) {
  ETLG_DYNAMIC_LDS(smem);
  plan_kernel<false, 1>(p, q, smem);
}

template <int RD>
__global__ __launch_bounds__(64, ETLG_PLAN_MINWAVES) void k_plan2(DecParams p, PlanParams q) {
  ETLG_DYNAMIC_LDS(smem);
  plan_kernel<true, RD>(p, q, smem);
}

}  // namespace etlg

extern "C" {

using namespace etlg;

void etlg_k_launch_plan_pre(const DecParams* p, const void* qv, hipStream_t s) {
  const PlanParams* q = (const PlanParams*)qv;
  hipLaunchKernelGGL(k_plan_pre, dim3((q->ntiles + (1u << kPreGroupLog) - 1u) >> kPreGroupLog), dim3(kPreWaves * 64), 0, s, *p, *q);
}

void etlg_k_launch_plan(const DecParams* p, const void* qv, hipStream_t s) {
  const PlanParams* q = (const PlanParams*)qv;
  // two tiles per wave whenever a row fits 8 dwords (its image then needs no LDS beyond the window); ETLG_PLAN_DBG bit 9 = one tile per wave
  if (q->lds_bytes == q->rows_off && q->max_row_dw <= 8u && !(q->dbg & 512u)) {
    if (q->max_row_dw <= 6u) hipLaunchKernelGGL(k_plan2<6>, dim3((q->ntiles + 1) / 2), dim3(64), q->lds_bytes, s, *p, *q);
    else hipLaunchKernelGGL(k_plan2<8>, dim3((q->ntiles + 1) / 2), dim3(64), q->lds_bytes, s, *p, *q);
  } else hipLaunchKernelGGL(k_plan, dim3(q->ntiles), dim3(64), q->lds_bytes, s, *p, *q);
}

int etlg_k_plan_set_lds(void) {
  const int a = hipFuncSetAttribute((const void*)k_plan, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) == hipSuccess ? 0 : 1;
  const int b = hipFuncSetAttribute((const void*)k_plan2<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) == hipSuccess ? 0 : 1;
  const int c = hipFuncSetAttribute((const void*)k_plan2<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) == hipSuccess ? 0 : 1;
  return a | b | c;
}

}  // extern "C"
