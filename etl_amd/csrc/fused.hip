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
#include "codec.hip.h"

namespace etlg {

constexpr unsigned long long ST_AGG = 1ull << 62, ST_INCL = 2ull << 62, ST_MASK = 3ull << 62;
constexpr uint32_t kMaxPolls = 1u << 22;

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

// Decoupled look-back executed by ONE wave (all 64 lanes call it). Returns the
// exclusive prefix of `agg` over tiles [0, tile) combined with `carry`.
template <class Op>
DEV uint64_t lookback(unsigned long long* desc, uint32_t tile, uint64_t agg, uint64_t carry, uint32_t* fail) {
  const int lane = threadIdx.x & 63;
  if (fail == nullptr) return carry;  // ablation only
  if (tile == 0) {
    if (lane == 0) __hip_atomic_store(&desc[0], ST_INCL | Op::f(carry, agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return carry;
  }
  if (lane == 0) __hip_atomic_store(&desc[tile], ST_AGG | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint64_t acc = Op::id();  // fold of tiles (base, tile) so far, newest part
  int64_t base = (int64_t)tile - 1;
  uint32_t polls = 0;
  for (;;) {
    const int64_t idx = base - lane;
    unsigned long long w = ST_INCL | carry;  // virtual tile -1: the carried state
    if (idx >= 0) w = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (idx < -1) w = ST_INCL | Op::id();
    const unsigned long long st = w & ST_MASK;
    const unsigned long long m_incl = __ballot(st == ST_INCL);
    const unsigned long long m_empty = __ballot(st == 0);
    const int first_incl = m_incl ? __builtin_ctzll(m_incl) : 64;
    const unsigned long long needed = first_incl >= 63 ? ~0ull : ((2ull << first_incl) - 1);
    if (m_empty & needed) {
      if (++polls > kMaxPolls) { if (lane == 0) atomicOr(fail, 1u); return Op::id(); }
      __builtin_amdgcn_s_sleep(2);
      continue;
    }
    // ordered fold of lanes [0, last] (higher lane = older tile): older ⊕ newer
    const int last = first_incl < 64 ? first_incl : 63;
    uint64_t v = lane <= last ? (uint64_t)(w & ~ST_MASK) : Op::id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t t = __shfl_down(v, d, 64);
      if (lane + d < 64) v = Op::f(t, v);
    }
    const uint64_t window = __shfl(v, 0, 64);
    acc = Op::f(window, acc);
    if (first_incl < 64) break;
    base -= 64;
  }
  if (lane == 0) __hip_atomic_store(&desc[tile], ST_INCL | Op::f(acc, agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return acc;
}

DEV uint32_t seg_pack30(uint32_t seg) { return ((seg >> 31) << 29) | (seg & 0x1FFFFFFFu); }
DEV uint32_t seg_unpack30(uint32_t s30) { return ((s30 >> 29) << 31) | (s30 & 0x1FFFFFFFu); }

// Everything after staging; `base` points at byte `win0` of the input (LDS or global).
template <int BLK>
DEV void tile_body(const DecParams& p, const FusedParams& q, uint32_t tile, uint32_t nt, const uint32_t* s_offs,
                   const u8* base, uint32_t win0, uint32_t* s32, uint64_t* s64) {
  const uint32_t tid = threadIdx.x;
  const bool live = tid < nt;
  const uint32_t f = tile * BLK + tid;
  // ---- phase 1: envelope, tag, structure
  FrameView v{f, 0, base, base};
  RowMsg m;
  bool wire_ok = true;
  if (live) {
    const uint32_t o0 = s_offs[tid], o1 = s_offs[tid + 1];
    if (o1 > o0 && o1 <= p.in_len) {
      v.fr = base + (o0 - win0);
      v.e = base + (o1 - win0);
      v.tag = classify_ptr(v.fr, o1 - o0);
    }
    wire_ok = frame_structure(v, m);
  }
  uint32_t cnt = 0, mark = 0;
  if (live) {
    if (consumes_ordinal(v.tag)) cnt = 1;
    if (v.tag == 'B') { cnt |= 0x80000000u; mark = ((f + 1) << 1) | 1; }
    if (v.tag == 'C') mark = (f + 1) << 1;
  }
  uint32_t tot_cnt, tot_mark;
  const uint32_t ic = block_scan_incl<2>(cnt, s32, &tot_cnt);
  const uint32_t im = block_scan_incl<1>(mark, s32 + 4, &tot_mark);
  uint32_t pm = __shfl_up(im, 1, 64);
  if ((tid & 63) == 63) s32[8 + (tid >> 6)] = im;
  __syncthreads();
  if ((tid & 63) == 0) pm = tid ? s32[8 + (tid >> 6) - 1] : 0;
  // ---- look-back 1: transaction state
  if (tid < 64) {
    const uint64_t agg = ((uint64_t)seg_pack30(tot_cnt) << 32) | tot_mark;
    const uint64_t carry = (uint64_t)(p.in_txn ? 1u : 0u);  // virtual Begin before frame 0; seg identity
    const uint64_t ex = lookback<OpTxn>(q.d_txn, tile, agg, carry, (q.dbg & 4) ? nullptr : &p.res->fused_fail);
    if (tid == 0) { s32[12] = seg_unpack30((uint32_t)(ex >> 32)); s32[13] = (uint32_t)ex; }
  }
  __syncthreads();
  const uint32_t bc = s32[12], bm = s32[13];
  const uint32_t seg = seg_combine(bc, ic);
  const uint32_t last = bm > pm ? bm : pm;
  TxnCtx tx;
  tx.in_txn = (last & 1u) != 0;
  tx.final_lsn = 0;
  if (tx.in_txn) tx.final_lsn = last == 1u ? p.final_lsn : ld_be64(p.in + p.offs[(last >> 1) - 1] + kBodyOff);
  {
    const uint64_t c = seg & 0x7FFFFFFFu;
    tx.ord = (seg & 0x80000000u) ? c - 1 : p.next_ord + c - 1;
  }
  // ---- phase 2: sizes
  uint32_t emit = 0, fixed = 0, heap = 0;
  uint64_t pay[3] = {0, 0, 0};
  int row_slot = -1;
  if (live) size_frame(p, v, tx, wire_ok, m, emit, fixed, heap, pay, row_slot);
  uint64_t tot_ev, tot_fx, tot_hp;
  const uint64_t x_ev = block_scan_excl64(emit, s64, &tot_ev);
  const uint64_t x_fx = block_scan_excl64(fixed, s64, &tot_fx);
  const uint64_t x_hp = block_scan_excl64(heap, s64, &tot_hp);
  const uint64_t p0 = block_sum64(pay[0], s64), p1 = block_sum64(pay[1], s64), p2 = block_sum64(pay[2], s64);
  // ---- look-back 2: output positions
  if (tid < 64) {
    const uint64_t a = lookback<OpAdd2>(q.d_outa, tile, (tot_ev << 32) | (uint32_t)(tot_hp >> 2), 0, (q.dbg & 4) ? nullptr : &p.res->fused_fail);
    const uint64_t b = lookback<OpAdd>(q.d_outb, tile, tot_fx >> 2, 0, (q.dbg & 4) ? nullptr : &p.res->fused_fail);
    if (tid == 0) { s64[4] = a; s64[5] = b; }
  }
  __syncthreads();
  const uint64_t pre_ev = s64[4] >> 32, pre_hp = (uint64_t)(uint32_t)s64[4] << 2, pre_fx = s64[5] << 2;
  if (tid == 0) {
    if (p0) atomicAdd((unsigned long long*)&p.res->payload[0], (unsigned long long)p0);
    if (p1) atomicAdd((unsigned long long*)&p.res->payload[1], (unsigned long long)p1);
    if (p2) atomicAdd((unsigned long long*)&p.res->payload[2], (unsigned long long)p2);
    if (tile == q.ntiles - 1) {  // the last tile knows the totals and the carried transaction state
      DevResult* r = p.res;
      r->n_events = pre_ev + tot_ev; r->fixed_bytes = pre_fx + tot_fx; r->heap_bytes = pre_hp + tot_hp;
      r->n_frames = p.nframes;
      const uint32_t sg = seg_combine(bc, tot_cnt);
      const uint32_t lm = bm > tot_mark ? bm : tot_mark;
      const bool it = (lm & 1u) != 0;
      r->out_in_txn = it;
      r->out_final_lsn = it ? (lm == 1u ? p.final_lsn : ld_be64(p.in + p.offs[(lm >> 1) - 1] + kBodyOff)) : 0;
      const uint64_t c = sg & 0x7FFFFFFFu;
      r->out_next_ord = (sg & 0x80000000u) ? c : p.next_ord + c;
    }
  }
  // ---- phase 3: decode + write
  if (!emit) return;
  const uint64_t ev_idx = pre_ev + x_ev, fx_off = pre_fx + x_fx, hp_off = pre_hp + x_hp;
  if (fx_off + fixed > p.fixed_cap || hp_off + heap > p.heap_cap || hp_off + heap > 0xFFFFFFFFull) {
    record_error(p, f, RK_DECODE, ETLG_E_WIRE);
    return;
  }
  if (q.dbg & 2) return;
  write_frame(p, v, tx, m, row_slot, ev_idx, fx_off, hp_off);
}

template <int BLK>
__global__ __launch_bounds__(BLK) void k_fused(DecParams p, FusedParams q) {
  extern __shared__ __attribute__((aligned(16))) u8 smem[];
  __shared__ uint32_t s_offs[BLK + 1];
  __shared__ uint32_t s32[16];
  __shared__ uint64_t s64[8];
  const uint32_t tid = threadIdx.x;
  if (tid == 0) s32[15] = atomicAdd(q.ticket, 1u);
  __syncthreads();
  const uint32_t tile = s32[15];
  if (tile >= q.ntiles) return;
  const uint32_t f0 = tile * BLK;
  const uint32_t nt = p.nframes - f0 < (uint32_t)BLK ? p.nframes - f0 : (uint32_t)BLK;
  for (uint32_t i = tid; i <= nt; i += BLK) s_offs[i] = p.offs[f0 + i];
  __syncthreads();
  const uint32_t span0 = s_offs[0], span1 = s_offs[nt];
  // every well-formed frame of the tile must lie inside [span0, span1) to be staged
  bool lane_ok = true;
  if (tid < nt) {
    const uint32_t o0 = s_offs[tid], o1 = s_offs[tid + 1];
    lane_ok = o1 <= o0 || o1 > p.in_len || (o0 >= span0 && o1 <= span1);
  }
  const uint32_t a0 = span0 & ~15u;
  const bool window_ok = q.in_aligned && span1 > span0 && span1 <= p.in_len && (uint64_t)(span1 - a0) + 16 <= q.lds_bytes;
  const bool use_lds = __syncthreads_and(lane_ok ? 1 : 0) && window_ok && !(q.dbg & 1);
  if (use_lds) {
    // coalesced staging: 16 B per lane per step; the tail that would cross in_len goes bytewise
    const uint32_t full_end = a0 + ((span1 - a0) & ~15u);  // last full 16-byte chunk boundary <= span1
    for (uint32_t c = a0 + 16 * tid; c < full_end; c += 16 * BLK)
      *(uint4*)(smem + (c - a0)) = *(const uint4*)(p.in + c);
    for (uint32_t c = full_end + tid; c < span1; c += BLK) smem[c - a0] = p.in[c];
    __syncthreads();
    tile_body<BLK>(p, q, tile, nt, s_offs, smem, a0, s32, s64);
  } else {
    tile_body<BLK>(p, q, tile, nt, s_offs, p.in, 0, s32, s64);
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
