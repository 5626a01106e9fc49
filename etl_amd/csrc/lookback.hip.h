// Two-level decoupled look-back shared by the single-pass kernels (fused.hip, cells.hip).
#pragma once
#include "codec.hip.h"

namespace etlg {

constexpr unsigned long long ST_AGG = 1ull << 62, ST_INCL = 2ull << 62, ST_MASK = 3ull << 62;
constexpr uint32_t kMaxPolls = 1u << 20;

// payload combiners (62-bit payloads, `a` older than `b`)
struct OpTxn {
  DEV static uint64_t id() { return 0; }
  DEV static uint64_t f(uint64_t a, uint64_t b) {
    const uint32_t sa = (uint32_t)(a >> 32), sb = (uint32_t)(b >> 32);  // flag at bit 29, count in bits 0..28
    const uint32_t ma = (uint32_t)a, mb = (uint32_t)b;
    const uint32_t s = (sb & (1u << 29)) ? sb : ((sa & (1u << 29)) | ((sa + sb) & 0x1FFFFFFFu));
    return ((uint64_t)s << 32) | (ma > mb ? ma : mb);
  }
};
struct OpAdd2 {  // two packed counters: hi 30 bits, lo 32 bits
  DEV static uint64_t id() { return 0; }
  DEV static uint64_t f(uint64_t a, uint64_t b) {
    return ((((a >> 32) + (b >> 32)) & 0x3FFFFFFFull) << 32) | (uint32_t)((uint32_t)a + (uint32_t)b);
  }
};
struct OpAdd {
  DEV static uint64_t id() { return 0; }
  DEV static uint64_t f(uint64_t a, uint64_t b) { return (a + b) & ~ST_MASK; }
};

// Two-level decoupled look-back, executed by ONE wave (all 64 lanes call it).
//
// Level 0: every tile publishes its aggregate in desc[tile] (status AGG only).
// Level 1: tiles are grouped by 64; the last tile of a group folds the group's 64
//          aggregates into gdesc[group] (AGG, then INCL once its own prefix is known).
// A tile's exclusive prefix = (prefix before its group, from one window over the group
// descriptors) ⊕ (fold of the earlier tiles of its own group, one window over desc).
// Both windows are independent of how far the predecessors have progressed beyond
// publishing their aggregate, so a batch whose tiles all start in lock-step (a grid of a
// few thousand tiles is only ~3 rounds of the chip) resolves in ~3 memory round trips
// instead of a 64-tiles-per-round-trip wavefront.
//
// Every word carries status + payload in ONE 64-bit value (relaxed agent-scope atomic
// store / load), so no fence is needed; nothing depends on placement or dispatch order
// beyond "a tile's predecessors have started" (ticket order); all spins are bounded.
// The two halves can be called apart (plan.hip): a tile publishes its aggregate as soon as it knows it, does work that
// does not need the prefix, and resolves afterwards — by then its predecessors' words have usually arrived.
DEV void lookback_publish(unsigned long long* desc, uint32_t tile, uint64_t agg) {
  if ((threadIdx.x & 63) == 0) __hip_atomic_store(&desc[tile], ST_AGG | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The first words a resolve looks at, requested ahead of time (a wave that resolves two tiles back to back asks for both tiles'
// words before it waits for the first: their round trips overlap). A word that has not been published yet is simply polled again.
struct LookbackPre { unsigned long long w0 = 0, w1 = 0; bool valid = false; };
DEV LookbackPre lookback_prefetch(unsigned long long* desc, unsigned long long* gdesc, uint32_t tile) {
  const int lane = threadIdx.x & 63;
  const uint32_t g = tile >> 6, j = tile & 63;
  LookbackPre r;
  r.valid = true;
  if ((uint32_t)lane < j) r.w0 = __hip_atomic_load(&desc[(g << 6) + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int64_t idx = (int64_t)g - 1 - 63 + lane;
  if (idx >= 0) r.w1 = __hip_atomic_load(&gdesc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return r;
}
template <class Op>
DEV uint64_t lookback_resolve(unsigned long long* desc, unsigned long long* gdesc, uint32_t tile, uint64_t agg, uint64_t carry, uint32_t* fail,
                              const LookbackPre& pre = LookbackPre());

template <class Op>
DEV uint64_t lookback(unsigned long long* desc, unsigned long long* gdesc, uint32_t tile, uint64_t agg,
                      uint64_t carry, uint32_t* fail) {
  if (fail == nullptr) return carry;  // ablation only
  lookback_publish(desc, tile, agg);
  return lookback_resolve<Op>(desc, gdesc, tile, agg, carry, fail);
}

template <class Op>
DEV uint64_t lookback_resolve(unsigned long long* desc, unsigned long long* gdesc, uint32_t tile, uint64_t agg,
                              uint64_t carry, uint32_t* fail, const LookbackPre& pre) {
  const int lane = threadIdx.x & 63;
  const uint32_t g = tile >> 6, j = tile & 63;
  uint32_t polls = 0;
  // the first window of group descriptors is requested now, so that its round trip overlaps window 0's
  unsigned long long w1_pre = pre.w1;
  if (!pre.valid) {
    const int64_t idx = (int64_t)g - 1 - 63 + lane;  // same lane mapping as window 1 below
    if (idx >= 0) w1_pre = __hip_atomic_load(&gdesc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  bool pre_valid = true;
  // ---- window 0: earlier tiles of this group (lane l <-> tile g*64 + l, l < j)
  unsigned long long w0 = pre.valid ? pre.w0 : 0ull;
  bool have = (uint32_t)lane >= j || (w0 & ST_MASK) != 0;  // lanes >= j have nothing to fetch
  for (;;) {
    if (!have) {
      w0 = __hip_atomic_load(&desc[(g << 6) + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      have = (w0 & ST_MASK) != 0;
    }
    if (!__ballot(!have)) break;
    if (++polls > kMaxPolls) { if (lane == 0) atomicOr(fail, 1u); return Op::id(); }
    __builtin_amdgcn_s_sleep(2);
  }
  // ordered fold, lower lane = older: inclusive scan then take lane j-1
  uint64_t v0 = (uint32_t)lane < j ? (uint64_t)(w0 & ~ST_MASK) : Op::id();
  // DPP ladder (see wave_scan_incl in codec.hip.h) on the two halves of the 64-bit payload; Op::id() == 0
#define ETLG_LB_DPP(ctrl, rmask) { \
    const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v0, ctrl, rmask, 0xF, false); \
    const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v0 >> 32), ctrl, rmask, 0xF, false); \
    v0 = Op::f(((uint64_t)hi_ << 32) | lo_, v0); }
  ETLG_LB_DPP(0x111, 0xF) ETLG_LB_DPP(0x112, 0xF) ETLG_LB_DPP(0x114, 0xF) ETLG_LB_DPP(0x118, 0xF) ETLG_LB_DPP(0x142, 0xA) ETLG_LB_DPP(0x143, 0xC)
#undef ETLG_LB_DPP
  const uint64_t local = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v0 >> 32), 63) << 32) |
                         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v0, 63);  // lanes >= j hold the identity, so lane 63 = fold of [0, j)
  // the last tile of a full group publishes the group aggregate
  const uint64_t group_agg = Op::f(local, agg);
  if (j == 63 && lane == 0) __hip_atomic_store(&gdesc[g], ST_AGG | group_agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // ---- window(s) 1: group descriptors before g, 64 at a time, OLDER groups in LOWER lanes (lane 63 <-> group
  //      `base`, lane l <-> group base - 63 + l), virtual group -1 = the carry. The fold runs from the newest
  //      inclusive descriptor of the window to lane 63 with the same DPP ladder as above.
  uint64_t acc = Op::id();
  int64_t base = (int64_t)g - 1;
  for (;;) {
    const int64_t idx = base - 63 + lane;
    unsigned long long w = ST_INCL | carry;
    if (idx >= 0) w = (pre_valid && idx == (int64_t)g - 1 - (63 - lane)) ? w1_pre : __hip_atomic_load(&gdesc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (idx < -1) w = ST_INCL | Op::id();
    pre_valid = false;
    const unsigned long long st = w & ST_MASK;
    const unsigned long long m_incl = __ballot(st == ST_INCL);
    const unsigned long long m_empty = __ballot(st == 0);
    const int first_incl = m_incl ? 63 - __builtin_clzll(m_incl) : -1;  // the newest inclusive descriptor of the window
    const unsigned long long needed = first_incl <= 0 ? ~0ull : (~0ull << first_incl);
    if (m_empty & needed) {
      if (++polls > kMaxPolls) { if (lane == 0) atomicOr(fail, 1u); return Op::id(); }
      __builtin_amdgcn_s_sleep(2);
      continue;
    }
    uint64_t v = lane >= (first_incl < 0 ? 0 : first_incl) ? (uint64_t)(w & ~ST_MASK) : Op::id();
#define ETLG_LB_DPP(ctrl, rmask) { \
    const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, ctrl, rmask, 0xF, false); \
    const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), ctrl, rmask, 0xF, false); \
    v = Op::f(((uint64_t)hi_ << 32) | lo_, v); }
    ETLG_LB_DPP(0x111, 0xF) ETLG_LB_DPP(0x112, 0xF) ETLG_LB_DPP(0x114, 0xF) ETLG_LB_DPP(0x118, 0xF) ETLG_LB_DPP(0x142, 0xA) ETLG_LB_DPP(0x143, 0xC)
#undef ETLG_LB_DPP
    const uint64_t wfold = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63) << 32) |
                           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
    acc = Op::f(wfold, acc);
    if (first_incl >= 0) break;
    base -= 64;
  }
  if (j == 63 && lane == 0) __hip_atomic_store(&gdesc[g], ST_INCL | Op::f(acc, group_agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return Op::f(acc, local);
}

// phase clocks are sampled (1 tile in 16): thousands of same-address atomics would distort what they measure
#ifndef ETLG_DBG_WORD
#define ETLG_DBG_WORD q.dbg
#endif
#ifndef ETLG_TSTAMP_WHO
#define ETLG_TSTAMP_WHO (threadIdx.x == 0)
#endif
#define TSTAMP(k) do { if ((ETLG_DBG_WORD & 8) && ETLG_TSTAMP_WHO && (blockIdx.x & 15) == 3) { const unsigned long long _t = clock64(); atomicAdd(&p.res->dbg_t[k], _t - s64[7]); s64[7] = _t; } } while (0)

DEV uint32_t seg_pack30(uint32_t seg) { return ((seg >> 31) << 29) | (seg & 0x1FFFFFFFu); }
DEV uint32_t seg_unpack30(uint32_t s30) { return ((s30 >> 29) << 31) | (s30 & 0x1FFFFFFFu); }

// In the fused kernel the transaction mark of a B / C frame carries the frame's BYTE
// offset ((o0 + 1) << 1 | isBegin, batches < 2 GiB), so the Begin's final_lsn is one load away.
DEV uint64_t final_lsn_of_mark(const DecParams& p, uint32_t mark) {
  return mark == 1u ? p.final_lsn : ld_be64(p.in + ((mark >> 1) - 1) + kBodyOff);
}

// The transaction look-back of the generic kernels (k_fused, k_cells), one wave. The state carried into the batch is NOT part of
// the fold (virtual group -1 holds the identity): it is patched in afterwards, and only a tile whose prefix needs it — no Begin /
// Commit before the tile in this batch, or no Begin (ordinals then continue the carried count) — reads it. With DecParams.flags
// bit 4 the batch before this one may still be running on the other decode stream: such a tile then waits for that batch's last
// tile to have written its totals (carry_ready); every other tile of the batch never looks at the predecessor at all. Results
// for the workgroup, written by the wave's first lane: s32[12] segment count, s32[13] mark, s64[6] final_lsn of the Begin the
// mark names (0 when it names none), s64[3] the ordinal the batch starts from. The same four values come back in registers (wave
// uniform) for a caller whose wave goes on with them at once: reading the LDS slots back without a barrier in between leaves the
// order of lane 0's stores and the other lanes' loads to the compiler (round 4: k_cells' sequential look-back read stale slots on the
// MI355X; a workgroup barrier sits between the stores and the loads everywhere else).
struct TxnStart { uint32_t seg, mark; uint64_t lsn, ord; };
DEV TxnStart txn_lookback(const DecParams& pg, unsigned long long* d_txn, uint32_t ntiles, uint32_t tile, uint64_t txn_agg, uint32_t* fail,
                      uint32_t* s32, uint64_t* s64) {
  const uint64_t ex = lookback<OpTxn>(d_txn, d_txn + ntiles, tile, txn_agg, 0ull, fail);
  const uint32_t seg = seg_unpack30((uint32_t)(ex >> 32));
  uint32_t mark = (uint32_t)ex;
  uint32_t in_txn = pg.in_txn;
  uint64_t final_lsn = pg.final_lsn, next_ord = pg.next_ord;
  if ((pg.flags & 16u) && pg.carry && (mark == 0u || !(seg & 0x80000000u))) {   // wave-uniform
    for (uint32_t polls = 0;; polls++) {
      if (__hip_atomic_load(&pg.carry->carry_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) break;
      if (polls > (1u << 15)) { if ((threadIdx.x & 63) == 0) atomicOr(fail, 1u); break; }
      __builtin_amdgcn_s_sleep(8);
    }
    in_txn = __hip_atomic_load(&pg.carry->out_in_txn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    final_lsn = __hip_atomic_load(&pg.carry->out_final_lsn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    next_ord = __hip_atomic_load(&pg.carry->out_next_ord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (mark == 0u && in_txn) mark = 1u;   // virtual Begin before frame 0
  TxnStart t;
  t.seg = seg; t.mark = mark; t.ord = next_ord;
  t.lsn = (mark & 1u) ? (mark == 1u ? final_lsn : ld_be64(pg.in + ((mark >> 1) - 1) + kBodyOff)) : 0ull;   // (one address for the wave)
  if ((threadIdx.x & 63) == 0) { s32[12] = t.seg; s32[13] = t.mark; s64[6] = t.lsn; s64[3] = t.ord; }
  ETLG_WAVE_JOIN();
  return t;
}
// the last tile of a single-pass kernel has written the batch's totals: the batch behind it may be waiting for them
DEV void carry_publish(DevResult* r) { __hip_atomic_store(&r->carry_ready, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

// Variant kernel head (k_fused, k_cells): the four side tables are read as ONE concatenation, up to four dwords per lane held in
// registers (`side_load`), and stored to LDS only after the tile's staging loads have been issued (`side_store`), so the whole
// copy costs one global round trip that overlaps the span loads instead of one round trip per table before anything else starts.
// NT = threads per workgroup. `side_load` also redirects the side-table pointers of `p` to the LDS copy.
struct SideRegs { uint32_t v0, v1, v2, v3, tot4; };
template <int NT>
DEV void side_load(DecParams& p, bool have_side, uint32_t* side_lds, uint32_t tid, SideRegs& r) {
  const uint32_t e1 = p.n_tables * (sizeof(DevTable) / 4), e2 = e1 + p.n_epochs * (sizeof(DevEpoch) / 4);
  const uint32_t e3 = e2 + p.n_slots * (sizeof(DevSlot) / 4), tot4 = have_side ? e3 + p.n_cols * (sizeof(DevCol) / 4) : 0u;
  const uint32_t* t0 = (const uint32_t*)p.tables; const uint32_t* t1 = (const uint32_t*)p.epochs;
  const uint32_t* t2 = (const uint32_t*)p.slots; const uint32_t* t3 = (const uint32_t*)p.cols;
  auto ld = [&](uint32_t j) { return j < e1 ? t0[j] : j < e2 ? t1[j - e1] : j < e3 ? t2[j - e2] : t3[j - e3]; };
  r.v0 = r.v1 = r.v2 = r.v3 = 0; r.tot4 = tot4;
  if (tid < tot4) r.v0 = ld(tid);
  if (tid + NT < tot4) r.v1 = ld(tid + NT);
  if (tid + 2 * NT < tot4) r.v2 = ld(tid + 2 * NT);
  if (tid + 3 * NT < tot4) r.v3 = ld(tid + 3 * NT);
  for (uint32_t j = tid + 4 * NT; j < tot4; j += NT) side_lds[j] = ld(j);  // side tables beyond four dwords per lane: the simple way
  if (have_side) {
    p.tables = (const DevTable*)side_lds; p.epochs = (const DevEpoch*)(side_lds + e1);
    p.slots = (const DevSlot*)(side_lds + e2); p.cols = (const DevCol*)(side_lds + e3);
  }
}
template <int NT>
DEV void side_store(uint32_t* side_lds, uint32_t tid, const SideRegs& r) {
  if (tid < r.tot4) side_lds[tid] = r.v0;
  if (tid + NT < r.tot4) side_lds[tid + NT] = r.v1;
  if (tid + 2 * NT < r.tot4) side_lds[tid + 2 * NT] = r.v2;
  if (tid + 3 * NT < r.tot4) side_lds[tid + 3 * NT] = r.v3;
}

}  // namespace etlg
