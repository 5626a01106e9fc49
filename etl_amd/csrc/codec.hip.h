// Device-side codec shared by the multi-pass kernels (kernels.hip) and the fused
// single-pass kernel (fused.hip): byte readers, workgroup scans, pgoutput message
// walkers, the text value codecs and the row writers. Everything is
// __forceinline__ and pointer based: called with a pointer derived from LDS the
// loads become ds_read, with a pointer into the input buffer global_load.
//
// Reference sections restated here (paths relative to the supabase/etl checkout):
//   wire grammar (A0)          postgres-replication 0.6.7 / PostgreSQL protocol docs
//   payload accounting (A3)    crates/etl/src/postgres/codec/event.rs:261-297
//   tuple -> row (A4/A5)       crates/etl/src/postgres/codec/event.rs:554-587, 938-983
//   value codecs (A6-A10)      crates/etl/src/postgres/codec/{text.rs:32-153,bool.rs:11-19,hex.rs:11-52,time.rs:75-161},
//                              crates/etl-postgres/src/{numeric.rs:108-458,time.rs:121-207}
//   update/delete (A11/A12)    crates/etl/src/postgres/codec/event.rs:437-527, 605-923
//   ownership / cache (A2/A14) crates/etl/src/replication/apply.rs:2836-2867, 3514-3519, 3705-3734
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/etlg.h"
#include "dev_types.h"
#include "float_fast.h"

namespace etlg {

typedef uint8_t u8;

#define DEV __device__ __forceinline__
// The declarations the SIMT test emulator (tests/simt/simt.h) has to substitute: reconvergence markers, out-of-line device
// functions defined in this header, and the dynamic LDS window of a kernel.
#ifndef DEV_NOINLINE
#define DEV_NOINLINE __device__ __attribute__((noinline))
#endif
#ifndef ETLG_SCALAR_COPY   // a wave-uniform value in a scalar register of its own (not a member of a spilled register tuple)
#define ETLG_SCALAR_COPY(dst, src) asm volatile("s_mov_b32 %0, %1" : "=s"(dst) : "s"(src))
#endif
#ifndef ETLG_WAVE_PRIO   // s_setprio: the issue arbiter of a SIMD prefers the wave with the higher value (0..3)
#define ETLG_WAVE_PRIO(n) __builtin_amdgcn_s_setprio(n)
#endif
#ifndef ETLG_WAVE_JOIN   // reconvergence point of a divergent region with wave collectives inside: nothing on the GPU
#define ETLG_WAVE_JOIN() ((void)0)
#endif
#ifndef ETLG_CONST_AS      // constant address space (scalar loads of wave-uniform descriptors); the emulator has one address space
#define ETLG_CONST_AS __attribute__((address_space(4)))
#endif
#ifndef ETLG_DYNAMIC_LDS
#define ETLG_DYNAMIC_LDS(name) extern __shared__ __attribute__((aligned(16))) u8 name[]
#endif
// LDS-DMA (plan.hip): every lane moves 16 bytes from its own global address to (wave-uniform LDS base) + 16 * lane; the data
// never passes through VGPRs. The wave waits for its pieces with ETLG_VMEM_WAIT before it reads the window.
#ifndef ETLG_GLDS16
#define ETLG_GLDS16(g, l) __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(g), (void __attribute__((address_space(3)))*)(l), 16, 0, 0)
#define ETLG_VMEM_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// One aligned ds_read_b32. Volatile keeps the compiler from fusing neighbouring dwords into a ds_read_b64 / b128 with 4-byte
// alignment, which the LDS serves at the misaligned rate (tools/ubench/lds_align.hip: 3.3x the cycles of an aligned read).
#define ETLG_LDS_AS __attribute__((address_space(3)))
#define ETLG_LDS_LD32(p) (*(volatile const ETLG_LDS_AS uint32_t*)(p))
#endif
// A pair of look-back words (plan.hip): ONE 16-byte load / store that bypasses the non-coherent caches (volatile: sc0 sc1), tracked by
// the compiler's own s_waitcnt bookkeeping (an inline-asm load would not be). Each 8-byte half validates itself (status bits),
// so nothing depends on the 16 bytes arriving as one.
#ifndef ETLG_LD_PAIR
typedef unsigned int etlg_v4u __attribute__((ext_vector_type(4)));
#define ETLG_LD_PAIR(ptr, a, l) do { const etlg_v4u t_ = *(const volatile __attribute__((address_space(1))) etlg_v4u*)(ptr); \
    (a) = ((unsigned long long)t_.y << 32) | t_.x; (l) = ((unsigned long long)t_.w << 32) | t_.z; } while (0)
#define ETLG_ST_PAIR(ptr, a, l) do { etlg_v4u t_; t_.x = (unsigned int)(a); t_.y = (unsigned int)((unsigned long long)(a) >> 32); t_.z = (unsigned int)(l); \
    t_.w = (unsigned int)((unsigned long long)(l) >> 32); *(volatile __attribute__((address_space(1))) etlg_v4u*)(ptr) = t_; } while (0)
#endif
#ifndef ETLG_PLAN_MINWAVES
#define ETLG_PLAN_MINWAVES 5   // k_plan: waves per SIMD the register allocator leaves room for (96 VGPRs; the ~8 KB LDS window per wave allows ~5)
#endif
#define DEV_DECODE DEV

// ------------------------------------------------------------- staging (k_fused, k_cells)
// Eight independent 16-byte loads per lane in flight before the first LDS store: one HBM round trip for a
// 113-byte-per-lane tile.
template <int NT>
DEV void stage_chunks(const uint8_t* in, uint8_t* stage, uint32_t a0, uint32_t full_end, uint32_t tid) {
  constexpr int W = 8;
  for (uint32_t c = a0 + 16 * tid; c < full_end; c += 16 * NT * W) {
    uint4 v[W];
#pragma unroll
    for (int k = 0; k < W; k++) {
      const uint32_t ck = c + 16 * NT * k;
      v[k] = make_uint4(0, 0, 0, 0);
      if (ck < full_end) v[k] = *(const uint4*)(in + ck);
    }
#pragma unroll
    for (int k = 0; k < W; k++) {
      const uint32_t ck = c + 16 * NT * k;
      if (ck < full_end) *(uint4*)(stage + (ck - a0)) = v[k];
    }
  }
}

// ------------------------------------------------------------- byte helpers
DEV uint32_t ld_be32(const u8* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return __builtin_bswap32(v); }
DEV uint64_t ld_be64(const u8* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return __builtin_bswap64(v); }
DEV uint32_t ld_be16(const u8* p) { return ((uint32_t)p[0] << 8) | p[1]; }
DEV uint32_t pad4(uint32_t n) { return (n + 3u) & ~3u; }

// Unaligned 4 / 8 byte loads built from ALIGNED dword loads + v_alignbyte. LDS serves a misaligned
// ds_read_b32 / b64 lane by lane (measured: ~1k cycles per wave-wide access with random alignment),
// while aligned dword reads run at full rate. These read up to the end of the last dword touched,
// i.e. at most 3 (ldu32) / 3 (ldu64) bytes beyond the value plus the alignment slack below it.
DEV uint32_t ldu32(const u8* p) {
  const uint32_t sh = (uint32_t)(uintptr_t)p & 3u;
  const uint32_t* q = (const uint32_t*)(p - sh);
  return __builtin_amdgcn_alignbyte(q[1], q[0], sh);
}
DEV uint64_t ldu64(const u8* p) {
  const uint32_t sh = (uint32_t)(uintptr_t)p & 3u;
  const uint32_t* q = (const uint32_t*)(p - sh);
  const uint32_t w0 = q[0], w1 = q[1], w2 = q[2];
  return __builtin_amdgcn_alignbyte(w1, w0, sh) | ((uint64_t)__builtin_amdgcn_alignbyte(w2, w1, sh) << 32);
}
DEV bool is_digit(uint32_t c) { return c - '0' < 10u; }
DEV uint32_t lower(uint32_t c) { return (c - 'A' < 26u) ? c + 32 : c; }

DEV void st64(uint32_t* w, uint64_t v) { w[0] = (uint32_t)v; w[1] = (uint32_t)(v >> 32); }

// Sequential byte reader. With `over` (the caller guarantees 7 readable bytes past any index it
// asks for: LDS-staged tiles keep 16 bytes of slack) it fetches 8 bytes per load, i.e. one memory
// round trip per 8 characters instead of one per character; without it every access is a byte load.
struct ByteWin {
  const u8* s; bool over;
  uint32_t base = 0x80000000u; uint64_t w = 0;
  DEV uint32_t at(uint32_t i) {  // ascending access
    if (!over) return s[i];
    uint32_t d = i - base;
    if (d >= 8u) { base = i; __builtin_memcpy(&w, s + i, 8); d = 0; }
    return (uint32_t)(w >> (8 * d)) & 0xFFu;
  }
  DEV uint32_t at_rev(uint32_t i) {  // descending access
    if (!over) return s[i];
    if (i - base >= 8u) { base = i >= 7u ? i - 7u : 0u; __builtin_memcpy(&w, s + base, 8); }
    return (uint32_t)(w >> (8 * (i - base))) & 0xFFu;
  }
};

DEV void record_error(const DecParams& p, uint32_t frame, uint32_t rank, uint32_t code) {
  unsigned long long key = ((unsigned long long)frame << 16) | ((unsigned long long)rank << 8) | code;
  atomicMin(&p.res->first_err, key);
}

// ETLG_F_ASYNC batches are chained on the device: the transaction state a batch starts from is what the batch before it
// left in its result block (stream order), not what the host knew when it enqueued this one.
// Returns false when that batch did not produce a result (an error, a plan that did not hold, a predecessor that failed in turn):
// this one must not run either — the host decodes both again, in order, when they are synced (DevResult.fused_fail bit 3).
DEV bool load_carry(DecParams& p) {
  if (!p.carry) return true;
  // (bit 1 alone — a fixed-width plan that gave its batch up over a shape it does not cover — is not a reason to stop: that kernel has
  // published the three words from the Begin / Commit frames it read. The host keeps this batch's result only if the second attempt at
  // that batch leaves the same words: finish_batch, `spare`; otherwise this batch is decoded again as it always was.)
  const bool ok = (p.carry->fused_fail & ~2u) == 0 && p.carry->first_err == kNoErr;
  p.in_txn = p.carry->out_in_txn; p.final_lsn = p.carry->out_final_lsn; p.next_ord = p.carry->out_next_ord;
  if (!ok && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&p.res->fused_fail, 8u);
  return ok;
}

// device slot index -> the slot id the arenas name (DevSlot.host_id); negative values (lookup failures) pass through
DEV uint32_t slot_host_id(const DecParams& p, int sl) { return sl >= 0 ? p.slots[sl].host_id : (uint32_t)sl; }

// ------------------------------------------------------------- wave scans
// Inclusive scan over the 64 lanes of a wave with DPP (row_shr 1/2/4/8 inside each row of 16, then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3): six VALU instructions, no LDS
// crossbar round trips (a __shfl_up ladder is six dependent ds_bpermute, ~100 cycles each).
// op(older, newer); `idn` is its left identity (lanes without a source read it).
#define ETLG_DPP(ctrl, rmask) { const uint32_t t_ = (uint32_t)__builtin_amdgcn_update_dpp((int)idn, (int)v, ctrl, rmask, 0xF, false); v = op(t_, v); }
template <class F>
DEV uint32_t wave_scan_incl(uint32_t v, F op, uint32_t idn) {
  ETLG_DPP(0x111, 0xF) ETLG_DPP(0x112, 0xF) ETLG_DPP(0x114, 0xF) ETLG_DPP(0x118, 0xF) ETLG_DPP(0x142, 0xA) ETLG_DPP(0x143, 0xC)
  return v;
}
#undef ETLG_DPP
DEV uint32_t wave_scan_add(uint32_t v) { return wave_scan_incl(v, [](uint32_t a, uint32_t b) { return a + b; }, 0u); }
DEV uint32_t wave_scan_max(uint32_t v) { return wave_scan_incl(v, [](uint32_t a, uint32_t b) { return a > b ? a : b; }, 0u); }
DEV uint32_t wave_last(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }  // lane 63 of an inclusive scan = the wave total

// ------------------------------------------------------------- block scans
DEV uint32_t seg_combine(uint32_t a, uint32_t b) {  // bit31 = "a Begin was seen", low bits = count since
  return (b & 0x80000000u) ? b : ((a & 0x80000000u) | ((a + b) & 0x7FFFFFFFu));
}

template <int OP>  // 0 sum, 1 max, 2 segmented count
DEV uint32_t op_apply(uint32_t a, uint32_t b) {
  if (OP == 0) return a + b;
  if (OP == 1) return a > b ? a : b;
  return seg_combine(a, b);
}

// Inclusive scan across the 256-lane workgroup. `lds` = 4 words of scratch.
// Returns the inclusive value; *total = workgroup aggregate.
template <int OP>
DEV uint32_t block_scan_incl(uint32_t v, uint32_t* lds, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v = op_apply<OP>(t, v);
  }
  if (lane == 63) lds[wave] = v;
  __syncthreads();
  uint32_t pre = 0;
  bool have = false;
  for (int w = 0; w < wave; w++) { pre = have ? op_apply<OP>(pre, lds[w]) : lds[w]; have = true; }
  if (have) v = op_apply<OP>(pre, v);
  if (total) {
    uint32_t t = lds[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) t = op_apply<OP>(t, lds[w]);
    *total = t;
  }
  __syncthreads();
  return v;
}

DEV uint64_t block_sum64(uint64_t v, uint64_t* lds) {  // returns the workgroup total
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  uint64_t t = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += lds[w];
  __syncthreads();
  return t;
}

// Exclusive 64-bit sum scan over the workgroup; *total = aggregate.
DEV uint64_t block_scan_excl64(uint64_t v, uint64_t* lds, uint64_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  uint64_t pre = 0, tot = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); w++) { if (w < wave) pre += lds[w]; tot += lds[w]; }
  __syncthreads();
  if (total) *total = tot;
  return pre + inc - v;
}

// Exclusive sum scan of three u32 values at once (one LDS exchange, two barriers).
// lds: 3 * nwaves words. tot[k] = workgroup totals.
DEV void block_scan3_excl(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t* lds, uint32_t tot[3]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
  const uint32_t ia = wave_scan_add(a), ib = wave_scan_add(b), ic = wave_scan_add(c);
  if (lane == 63) { lds[3 * wave] = ia; lds[3 * wave + 1] = ib; lds[3 * wave + 2] = ic; }
  __syncthreads();
  uint32_t pa = 0, pb = 0, pc = 0, ta = 0, tb = 0, tc = 0;
  for (int w = 0; w < nw; w++) {
    const uint32_t xa = lds[3 * w], xb = lds[3 * w + 1], xc = lds[3 * w + 2];
    if (w < wave) { pa += xa; pb += xb; pc += xc; }
    ta += xa; tb += xb; tc += xc;
  }
  __syncthreads();
  a = pa + ia - a; b = pb + ib - b; c = pc + ic - c;
  tot[0] = ta; tot[1] = tb; tot[2] = tc;
}

// Transaction scan: inclusive segmented count + EXCLUSIVE running max in one exchange.
// lds: 2 * nwaves words.
DEV void block_scan_txn(uint32_t cnt, uint32_t mark, uint32_t* lds, uint32_t& seg_incl, uint32_t& mark_excl,
                        uint32_t& tot_cnt, uint32_t& tot_mark) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
  const uint32_t ic = wave_scan_incl(cnt, [](uint32_t a, uint32_t b) { return seg_combine(a, b); }, 0u);
  const uint32_t im = wave_scan_max(mark);
  // exclusive running max: the previous lane's inclusive value (wave_shr:1, 0 into lane 0)
  uint32_t pm = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)im, 0x138, 0xF, 0xF, false);
  if (lane == 63) { lds[2 * wave] = ic; lds[2 * wave + 1] = im; }
  __syncthreads();
  uint32_t pc = 0, pmx = 0, tcn = 0, tmx = 0;
  for (int w = 0; w < nw; w++) {
    const uint32_t xc = lds[2 * w], xm = lds[2 * w + 1];
    if (w < wave) { pc = seg_combine(pc, xc); pmx = pmx > xm ? pmx : xm; }
    tcn = seg_combine(tcn, xc); tmx = tmx > xm ? tmx : xm;
  }
  __syncthreads();
  seg_incl = seg_combine(pc, ic);
  mark_excl = pmx > pm ? pmx : pm;
  tot_cnt = tcn; tot_mark = tmx;
}

// --------------------------------------------------------- frame geometry
// CopyData := 'd' Int32-BE(len incl. itself) payload
// payload  := 'w' u64 wal_start u64 wal_end i64 ts <pgoutput msg> | 'k' u64 i64 u8
constexpr uint32_t kHdr = 5;        // 'd' + length
constexpr uint32_t kTagOff = 30;    // offset of the pgoutput tag inside the frame
constexpr uint32_t kBodyOff = 31;   // first byte after the tag

DEV bool consumes_ordinal(uint32_t tag) {  // apply.rs:2284-2292, 2339, 2377, 2457, 2501, 2545, 2587
  return tag == 'B' || tag == 'C' || tag == 'R' || tag == 'I' || tag == 'U' || tag == 'D' || tag == 'T';
}

// ------------------------------------------------------- side-input lookups
DEV int find_table(const DecParams& p, uint32_t table_id) {
  int lo = 0, hi = (int)p.n_tables - 1;
  while (lo <= hi) {
    int mid = (lo + hi) >> 1;
    uint32_t v = p.tables[mid].table_id;
    if (v == table_id) return mid;
    if (v < table_id) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

// should_apply_changes: apply.rs:2626-2639 -> 2836-2867 / 3514-3519
DEV bool should_apply(const DecParams& p, int ti, uint32_t table_id, uint64_t remote_final_lsn) {
  if (p.worker_kind == ETLG_WORKER_TABLE_SYNC) return p.sync_table == table_id;
  if (ti < 0) return false;
  const DevTable& t = p.tables[ti];
  if (t.state_kind == ETLG_TS_READY) return true;
  if (t.state_kind == ETLG_TS_SYNC_DONE) return t.state_lsn <= remote_final_lsn;
  return false;
}

// get_replicated_table_schema (apply.rs:3705-3734) at stream position `f`:
// the shared-table-cache entry in force just before frame f.
// returns slot >= 0, or -ETLG_E_MISSING_SHARED_STATE / -ETLG_E_WAITING_RELATION
DEV int cache_slot_before(const DecParams& p, int ti, uint32_t f) {
  if (ti < 0) return -(int)ETLG_E_MISSING_SHARED_STATE;
  const DevTable& t = p.tables[ti];
  uint32_t kind = t.init_kind;
  int slot = t.init_slot;
  // epochs are few (one per R / DDL message of this table in the batch)
  uint32_t lo = t.ep_begin, hi = t.ep_end;  // first epoch with frame >= f
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (p.epochs[mid].frame < f) lo = mid + 1; else hi = mid;
  }
  if (lo > t.ep_begin) { kind = p.epochs[lo - 1].kind; slot = p.epochs[lo - 1].slot; }
  if (kind == 0) return -(int)ETLG_E_MISSING_SHARED_STATE;
  if (kind == 1) return -(int)ETLG_E_WAITING_RELATION;
  return slot;
}

// The epoch created by the R message at frame f itself (nullptr if the host did not
// process it, e.g. because it lies behind the host's own error frame).
DEV const DevEpoch* epoch_at(const DecParams& p, int ti, uint32_t f) {
  if (ti < 0) return nullptr;
  const DevTable& t = p.tables[ti];
  uint32_t lo = t.ep_begin, hi = t.ep_end;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (p.epochs[mid].frame < f) lo = mid + 1; else hi = mid;
  }
  if (lo < t.ep_end && p.epochs[lo].frame == f && p.epochs[lo].kind == 2) return &p.epochs[lo];
  return nullptr;
}

// ---------------------------------------------------------- message shapes
// Tuple := i16 ncols, ncols x { 'n' | 'u' | 't' i32 len bytes | 'b' i32 len bytes }
// Structural walk: bounds, tags, value bytes (calculate_tuple_bytes, codec/event.rs:261-271).
DEV bool walk_tuple(const u8*& c, const u8* e, uint32_t& ncols, uint32_t& vbytes, bool over = false) {
  if (e - c < 2) return false;
  const uint32_t n = ld_be16(c);
  c += 2;
  if (n & 0x8000u) return false;  // negative count
  ncols = n;
  for (uint32_t i = 0; i < n; i++) {
    if (c >= e) return false;
    // cell tag and length in one load where the bytes may be over-read (LDS-staged tiles)
    uint64_t head = 0;
    if (over) __builtin_memcpy(&head, c, 8);
    const uint32_t t = over ? (uint32_t)head & 0xFFu : (uint32_t)*c;
    c++;
    if (t == 'n' || t == 'u') continue;
    if (t != 't' && t != 'b') return false;
    if (e - c < 4) return false;
    const uint32_t len = over ? __builtin_bswap32((uint32_t)(head >> 8)) : ld_be32(c);
    c += 4;
    if ((len & 0x80000000u) || (uint64_t)(e - c) < len) return false;
    vbytes += len;
    c += len;
  }
  return true;
}

struct RowMsg {
  uint32_t rel_id;
  const u8* old_t;   // at the i16 column count of the old/key tuple (nullptr if none)
  const u8* new_t;   // likewise for the new tuple (nullptr for D)
  uint32_t old_kind; // ETLG_OLD_*
  uint32_t old_n, new_n;
  uint32_t vbytes;   // payload bytes (A3)
};

// I := u32 rel 'N' Tuple ; U := u32 rel ['K'|'O' Tuple] 'N' Tuple ; D := u32 rel ('K'|'O') Tuple
DEV bool parse_row_msg(uint32_t tag, const u8* b, const u8* e, RowMsg& m, bool over = false) {
  if (e - b < 5) return false;
  m.rel_id = ld_be32(b);
  const u8* c = b + 4;
  m.old_t = nullptr; m.new_t = nullptr; m.old_kind = ETLG_OLD_NONE; m.old_n = m.new_n = 0; m.vbytes = 0;
  uint32_t t = *c++;
  if (tag == 'I') {
    if (t != 'N') return false;
    m.new_t = c;
    return walk_tuple(c, e, m.new_n, m.vbytes, over);
  }
  if (tag == 'U') {
    if (t == 'K' || t == 'O') {
      m.old_kind = t == 'K' ? ETLG_OLD_KEY : ETLG_OLD_FULL;
      m.old_t = c;
      if (!walk_tuple(c, e, m.old_n, m.vbytes, over)) return false;
      if (c >= e) return false;
      t = *c++;
    }
    if (t != 'N') return false;
    m.new_t = c;
    return walk_tuple(c, e, m.new_n, m.vbytes, over);
  }
  // 'D'
  if (t != 'K' && t != 'O') return false;
  m.old_kind = t == 'K' ? ETLG_OLD_KEY : ETLG_OLD_FULL;
  m.old_t = c;
  return walk_tuple(c, e, m.old_n, m.vbytes, over);
}

DEV bool has_cstr(const u8*& c, const u8* e) {
  while (c < e) if (*c++ == 0) return true;
  return false;
}

// Cell iterator over a structurally valid tuple.
struct CellIt {
  const u8* c;
  DEV void begin(const u8* tuple) { c = tuple + 2; }
  DEV uint32_t next(const u8*& data, uint32_t& len) {
    const uint32_t t = *c++;
    data = nullptr; len = 0;
    if (t == 't' || t == 'b') { len = ld_be32(c); c += 4; data = c; c += len; }
    return t;
  }
};

// ----------------------------------------------------------------- UTF-8
// core::str::from_utf8 (strict RFC 3629), call site codec/event.rs:976.
DEV_NOINLINE bool utf8_valid(const u8* s, uint32_t n) {
  uint32_t i = 0;
  while (i < n) {
    // ASCII runs, 8 bytes at a time (re-entered after every multi-byte character)
    while (i + 8 <= n) {
      uint64_t w; __builtin_memcpy(&w, s + i, 8);
      if (w & 0x8080808080808080ull) break;
      i += 8;
    }
    if (i >= n) break;
    const uint32_t c = s[i];
    if (c < 0x80) { i++; continue; }
    if (c >= 0xC2 && c <= 0xDF) {
      if (i + 1 >= n || (s[i + 1] & 0xC0) != 0x80) return false;
      i += 2;
    } else if (c >= 0xE0 && c <= 0xEF) {
      if (i + 2 >= n) return false;
      const uint32_t c1 = s[i + 1], c2 = s[i + 2];
      if ((c1 & 0xC0) != 0x80 || (c2 & 0xC0) != 0x80) return false;
      if (c == 0xE0 && c1 < 0xA0) return false;
      if (c == 0xED && c1 > 0x9F) return false;
      i += 3;
    } else if (c >= 0xF0 && c <= 0xF4) {
      if (i + 3 >= n) return false;
      const uint32_t c1 = s[i + 1], c2 = s[i + 2], c3 = s[i + 3];
      if ((c1 & 0xC0) != 0x80 || (c2 & 0xC0) != 0x80 || (c3 & 0xC0) != 0x80) return false;
      if (c == 0xF0 && c1 < 0x90) return false;
      if (c == 0xF4 && c1 > 0x8F) return false;
      i += 4;
    } else {
      return false;
    }
  }
  return true;
}

// Unicode White_Space (what str::trim strips), numeric.rs:112.
DEV uint32_t ws_len_at(const u8* p, uint32_t n) {
  if (n == 0) return 0;
  const uint32_t c = p[0];
  if ((c >= 0x09 && c <= 0x0D) || c == 0x20) return 1;
  if (c == 0xC2 && n >= 2 && (p[1] == 0x85 || p[1] == 0xA0)) return 2;
  if (c == 0xE1 && n >= 3 && p[1] == 0x9A && p[2] == 0x80) return 3;
  if (c == 0xE2 && n >= 3) {
    if (p[1] == 0x80 && ((p[2] >= 0x80 && p[2] <= 0x8A) || p[2] == 0xA8 || p[2] == 0xA9 || p[2] == 0xAF)) return 3;
    if (p[1] == 0x81 && p[2] == 0x9F) return 3;
  }
  if (c == 0xE3 && n >= 3 && p[1] == 0x80 && p[2] == 0x80) return 3;
  return 0;
}
DEV void trim_ws(const u8*& s, uint32_t& n) {
  for (;;) { uint32_t l = ws_len_at(s, n); if (!l) break; s += l; n -= l; }
  for (;;) {
    if (!n) break;
    uint32_t i = n - 1;
    while (i > 0 && (s[i] & 0xC0) == 0x80) i--;
    uint32_t l = ws_len_at(s + i, n - i);
    if (l != n - i || l == 0) break;
    n = i;
  }
}
DEV bool ieq(const u8* s, uint32_t n, const char* lit, uint32_t ln) {
  if (n != ln) return false;
  for (uint32_t i = 0; i < n; i++) if (lower(s[i]) != (uint32_t)lit[i]) return false;
  return true;
}

// ------------------------------------------------------------ value codecs
// Rust `iN::from_str` / `u32::from_str` (codec/text.rs:40-51,135-138).
// Returns false on error. `bits`: 16/32/64; `is_signed`.
DEV_NOINLINE bool parse_int(const u8* s, uint32_t n, bool is_signed, int bits, int64_t& out) {
  if (n == 0) return false;
  bool neg = false;
  uint32_t i = 0;
  if (s[0] == '+' || s[0] == '-') {
    if (n == 1) return false;
    if (s[0] == '-') { if (!is_signed) return false; neg = true; }
    i = 1;
  }
  while (i < n && s[i] == '0') i++;      // leading zeros are legal and carry no magnitude
  if (n - i > 19) return false;          // every supported type overflows at 20 significant digits
  // <= 19 digits cannot overflow u64: accumulate 9 digits at a time in 32 bits, no division anywhere
  static const uint32_t p10[10] = {1u, 10u, 100u, 1000u, 10000u, 100000u, 1000000u, 10000000u, 100000000u, 1000000000u};
  uint64_t mag = 0;
  while (i < n) {
    uint32_t chunk = 0, k = 0;
    while (i < n && k < 9) {
      const uint32_t d = s[i] - '0';
      if (d > 9) return false;
      chunk = chunk * 10 + d;
      i++; k++;
    }
    mag = mag * p10[k] + chunk;
  }
  uint64_t lim;
  if (is_signed) lim = neg ? (1ull << (bits - 1)) : (1ull << (bits - 1)) - 1;
  else lim = bits == 64 ? ~0ull : (1ull << bits) - 1;
  if (mag > lim) return false;
  out = neg ? (int64_t)(0 - mag) : (int64_t)mag;
  return true;
}

// SWAR variant for callers that may read up to 7 bytes past the text (LDS-staged tiles keep 16
// bytes of slack): four ASCII digits per 32-bit word, no per-character loop and no 64-bit
// multiplies (those cost 4-6 VALU instructions each on this hardware; a 10-digit value is three
// words, ~45 instructions).
// `w` = 4 bytes, first digit in byte 0. Returns their value; clears `ok` on a non-digit.
DEV uint32_t digits4(uint32_t w, bool& ok) {
  const uint32_t t = w - 0x30303030u;
  ok = ok && ((((w + 0x46464646u) | t) & 0x80808080u) == 0);
  const uint32_t p = (t * 10u + (t >> 8)) & 0x00FF00FFu;  // two 2-digit numbers (bytes 0 and 2)
  return (p & 0xFFu) * 100u + (p >> 16);
}
// The leading r (1..4) digits of a text, as a 4-digit word left-padded with '0'.
DEV uint32_t lead_word(const u8* s, uint32_t r) {
  uint32_t w; __builtin_memcpy(&w, s, 4);
  return r == 4 ? w : (w << (8 * (4 - r))) | (0x30303030u >> (8 * r));
}
DEV bool swar_digits8(const u8* s, uint32_t n /* 1..8 */, uint32_t& out) {
  bool ok = true;
  if (n <= 4) { out = digits4(lead_word(s, n), ok); return ok; }
  uint32_t lo; __builtin_memcpy(&lo, s + (n - 4), 4);
  out = digits4(lead_word(s, n - 4), ok) * 10000u + digits4(lo, ok);
  return ok;
}
DEV bool parse_int_swar(const u8* s, uint32_t n, bool is_signed, int bits, int64_t& out) {
  if (n == 0) return false;
  bool neg = false;
  const uint32_t c0 = s[0];
  if (c0 == '+' || c0 == '-') {
    if (n == 1) return false;
    if (c0 == '-') { if (!is_signed) return false; neg = true; }
    s++; n--;
  }
  if (n > 19) {  // long (leading zeros): the out-of-line parser gets its own result slot, so that `out` never has its
                 // address taken and stays in registers on the fast path (it lived in scratch: one memory round trip per cell)
    int64_t slow = 0;
    const bool ok_slow = parse_int(neg || c0 == '+' ? s - 1 : s, neg || c0 == '+' ? n + 1 : n, is_signed, bits, slow);
    out = slow;
    return ok_slow;
  }
  // groups of 4 digits from the right; the leading group has r = 1..4 digits
  const uint32_t r = ((n - 1) & 3u) + 1u, ng = (n - r) >> 2;  // ng full groups after the leading one (0..4)
  bool ok = true;
  const uint32_t lead = digits4(lead_word(s, r), ok);
  const u8* g = s + r;
  uint64_t mag;
  if (ng == 0) mag = lead;
  else if (ng == 1) {  // up to 8 digits: 32-bit arithmetic
    uint32_t w; __builtin_memcpy(&w, g, 4);
    mag = lead * 10000u + digits4(w, ok);
  } else {
    // two groups per 8-byte load; the first accumulate is a 32 x 32 -> 64 multiply-add
    uint64_t x; __builtin_memcpy(&x, g, 8);
    mag = (uint64_t)lead * 100000000u + (digits4((uint32_t)x, ok) * 10000u + digits4((uint32_t)(x >> 32), ok));
    g += 8;
    if (ng >= 4) {
      __builtin_memcpy(&x, g, 8);
      mag = mag * 100000000ull + (digits4((uint32_t)x, ok) * 10000u + digits4((uint32_t)(x >> 32), ok));
      g += 8;
    }
    if (ng & 1u) {
      uint32_t w; __builtin_memcpy(&w, g, 4);
      mag = mag * 10000ull + digits4(w, ok);
    }
  }
  if (!ok) return false;
  uint64_t lim;
  if (is_signed) lim = neg ? (1ull << (bits - 1)) : (1ull << (bits - 1)) - 1;
  else lim = bits == 64 ? ~0ull : (1ull << bits) - 1;
  if (mag > lim) return false;
  out = neg ? (int64_t)(0 - mag) : (int64_t)mag;
  return true;
}

// PgNumeric::from_str (crates/etl-postgres/src/numeric.rs:108-135, 246-267,
// 276-396) + the group arithmetic of convert_to_base_10000 (:404-458).
struct NumShape {
  uint32_t kind;        // ETLG_NUM_*
  uint32_t sign;
  int32_t weight;
  uint32_t scale;
  uint32_t ngroups;
  int32_t offset;       // alignment of decimal digit 0 inside its base-10000 group
  int32_t first_group, last_group;
  const u8* mant;       // first mantissa byte (after sign / leading '.')
  uint32_t mant_len;    // mantissa bytes (digits, '.', '_')
  uint32_t dot_at;      // numeric_plain only: index of the '.' inside the mantissa, mant_len when there is none
};

// weight / first and last base-10000 group from the decimal weight of the first digit and the first / last non-zero digit
DEV bool numeric_groups(NumShape& o, int32_t dweight, int32_t first_nz, int32_t last_nz) {
  if (first_nz < 0) { o.sign = 0; o.weight = 0; return true; }  // canonical zero
  const int32_t weight = dweight >= 0 ? (dweight + 4) / 4 - 1 : -((-dweight - 1) / 4 + 1);
  o.offset = (weight + 1) * 4 - (dweight + 1);
  o.first_group = (o.offset + first_nz) / 4;
  o.last_group = (o.offset + last_nz) / 4;
  const int32_t fw = weight - o.first_group;
  if (fw < -32768 || fw > 32767) return false;
  o.weight = fw;
  o.ngroups = (uint32_t)(o.last_group - o.first_group + 1);
  return true;
}

DEV bool numeric_scan(const u8* s, uint32_t n, NumShape& o, bool over = false) {
  trim_ws(s, n);
  if (n == 0) return false;
  o.sign = 0; o.kind = ETLG_NUM_VALUE; o.weight = 0; o.scale = 0; o.ngroups = 0;
  bool explicit_sign = false;
  if (s[0] == '+') { s++; n--; explicit_sign = true; }
  else if (s[0] == '-') { o.sign = 1; s++; n--; explicit_sign = true; }
  if (!(n && (is_digit(s[0]) || s[0] == '.'))) {
    // parse_special_value; `rest.trim_end()` is a no-op after the outer trim
    if (ieq(s, n, "nan", 3)) { if (explicit_sign) return false; o.kind = ETLG_NUM_NAN; return true; }
    if (ieq(s, n, "infinity", 8) || ieq(s, n, "inf", 3)) {  // the sign lives in the variant, not in a field
      o.kind = o.sign ? ETLG_NUM_NINF : ETLG_NUM_PINF; o.sign = 0; return true;
    }
    return false;
  }
  // parse_numeric_value
  uint32_t pos = 0;
  ByteWin bw{s, over};
  bool have_dp = false;
  int32_t dweight = -1;
  uint32_t dscale = 0;
  if (bw.at(0) == '.') { have_dp = true; pos = 1; }
  if (!(pos < n && is_digit(bw.at(pos)))) return false;
  o.mant = s + pos;
  int32_t k = 0, first_nz = -1, last_nz = -1;  // decimal digit indexes
  while (pos < n) {
    const uint32_t c = bw.at(pos);
    if (is_digit(c)) {
      pos++;
      if (c != '0') { if (first_nz < 0) first_nz = k; last_nz = k; }
      k++;
      if (!have_dp) dweight++; else dscale++;
    } else if (c == '.') {
      if (have_dp) return false;
      have_dp = true; pos++;
      if (pos < n && bw.at(pos) == '_') return false;
    } else if (c == '_') {
      pos++;
      if (!(pos < n && is_digit(bw.at(pos)))) return false;
    } else break;
  }
  o.mant_len = (uint32_t)((s + pos) - o.mant);
  if (pos < n && (bw.at(pos) == 'e' || bw.at(pos) == 'E')) {
    pos++;
    int64_t ex = 0;
    bool exneg = false;
    if (pos < n && bw.at(pos) == '+') pos++;
    else if (pos < n && bw.at(pos) == '-') { exneg = true; pos++; }
    if (!(pos < n && is_digit(bw.at(pos)))) return false;
    while (pos < n) {
      const uint32_t c = bw.at(pos);
      if (is_digit(c)) {
        pos++;
        ex = ex * 10 + (c - '0');
        if (ex > 0x3FFFFFFF) return false;  // i32::MAX / 2 guard -> ValueOutOfRange
      } else if (c == '_') {
        pos++;
        if (!(pos < n && is_digit(bw.at(pos)))) return false;
      } else break;
    }
    if (exneg) ex = -ex;
    dweight += (int32_t)ex;
    int64_t ds = (int64_t)dscale - ex;
    dscale = ds < 0 ? 0u : (uint32_t)ds;
  }
  if (pos != n) return false;
  if (dscale > 16383) return false;
  o.scale = dscale;
  return numeric_groups(o, dweight, first_nz, last_nz);
}

// The same result for the texts Postgres itself prints for ordinary values — [+-] digits [. digits], at most 24 characters
// after the sign, nothing else (no exponent, '_', whitespace, NaN / Infinity) — without a character loop: the characters are
// classified four at a time and the positions of the '.' and of the first / last non-zero digit come out of bit masks.
// Returns false for every other text, valid or not: the caller then runs numeric_scan. Staged (LDS) text only: the loads
// may run up to 7 bytes past the text.
DEV bool numeric_plain(const u8* s, uint32_t n, NumShape& o) {
  if (n == 0) return false;
  const uint32_t c0 = s[0];
  const uint32_t start = (c0 == '+' || c0 == '-') ? 1u : 0u;
  const uint32_t m = n - start;
  if (m - 1u > 23u) return false;
  const u8* p = s + start;
  uint32_t dm = 0, pm = 0, nz = 0;  // bit i: character i is a digit / the '.' / a digit other than '0'
#pragma unroll
  for (uint32_t j = 0; j < 6; j++) {
    if (4 * j < m) {
      uint32_t x; __builtin_memcpy(&x, p + 4 * j, 4);
      const uint32_t t = (x & 0x7F7F7F7Fu) ^ 0x30303030u;                 // '0'..'9' -> 0..9
      const uint32_t dig = ~(t + 0x76767676u) & ~x & 0x80808080u;          // bit 7 of a byte: it is an ASCII digit
      const uint32_t z = x ^ 0x2E2E2E2Eu;
      const uint32_t dot = ~(((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;   // ... it is '.'
      const uint32_t nzd = dig & ((t + 0x7F7F7F7Fu) | t);                   // ... a digit and t != 0
      // the four bit-7 flags of a word -> 4 adjacent bits (the partial products of the multiply never collide)
      dm |= (((dig >> 7) * 0x00204081u) >> 21 & 0xFu) << (4 * j);
      pm |= (((dot >> 7) * 0x00204081u) >> 21 & 0xFu) << (4 * j);
      nz |= (((nzd >> 7) * 0x00204081u) >> 21 & 0xFu) << (4 * j);
    }
  }
  const uint32_t vm = (1u << m) - 1u;   // m <= 24
  dm &= vm; pm &= vm; nz &= vm;
  if ((dm | pm) != vm || (pm & (pm - 1u)) != 0 || (pm == 1u && m < 2)) return false;
  const uint32_t dp = pm ? (uint32_t)__builtin_ctz(pm) : m;   // digits in front of the '.'
  const bool lead_dot = dp == 0;
  o.sign = c0 == '-' ? 1u : 0u; o.kind = ETLG_NUM_VALUE; o.weight = 0; o.ngroups = 0;
  o.scale = pm ? m - dp - 1 : 0u;
  o.mant = p + (lead_dot ? 1 : 0);
  o.mant_len = m - (lead_dot ? 1u : 0u);
  o.dot_at = (pm && !lead_dot) ? dp : o.mant_len;
  int32_t first_nz = -1, last_nz = -1;   // as indexes of decimal digits
  if (nz) {
    const uint32_t fb = (uint32_t)__builtin_ctz(nz), lb = 31u - (uint32_t)__builtin_clz(nz);
    first_nz = (int32_t)(fb - (fb > dp ? 1u : 0u));
    last_nz = (int32_t)(lb - (lb > dp ? 1u : 0u));
  }
  return numeric_groups(o, (int32_t)dp - 1, first_nz, last_nz);
}

// Writes the etlg_numeric_hdr + digits at `dst` (4-byte aligned); returns bytes incl. padding.
DEV uint32_t numeric_emit(const NumShape& o, u8* dst, bool over = false) {
  uint32_t* w = (uint32_t*)dst;
  w[0] = o.kind | (o.sign << 8) | ((uint32_t)(uint16_t)(int16_t)o.weight << 16);
  w[1] = o.scale | ((o.ngroups & 0xFFFFu) << 16);
  uint16_t* dg = (uint16_t*)(dst + 8);
  if (o.ngroups) {
    int32_t k = 0;
    int32_t cur = o.first_group;
    uint32_t acc = 0;
    static const uint32_t p10[4] = {1000, 100, 10, 1};
    ByteWin bw{o.mant, over};
    for (uint32_t i = 0; i < o.mant_len; i++) {
      const uint32_t c = bw.at(i);
      if (!is_digit(c)) continue;
      const int32_t pos = o.offset + k;
      k++;
      const int32_t g = pos >> 2;
      if (g < o.first_group) continue;
      if (g > o.last_group) break;
      while (cur < g) { dg[cur - o.first_group] = (uint16_t)acc; acc = 0; cur++; }
      acc += (c - '0') * p10[pos & 3];
    }
    while (cur <= o.last_group) { dg[cur - o.first_group] = (uint16_t)acc; acc = 0; cur++; }
    if (o.ngroups & 1) dg[o.ngroups] = 0;  // zero padding up to 4 bytes
  }
  return pad4(8 + 2 * o.ngroups);
}

// numeric_emit for a shape numeric_plain produced: digit k of the mantissa is byte k (+ 1 behind the '.'), so every group is
// four independent byte reads; two groups leave as one dword.
DEV uint32_t numeric_emit_plain(const NumShape& o, u8* dst) {
  uint32_t* w = (uint32_t*)dst;
  w[0] = o.kind | (o.sign << 8) | ((uint32_t)(uint16_t)(int16_t)o.weight << 16);
  w[1] = o.scale | ((o.ngroups & 0xFFFFu) << 16);
  const int32_t nd = (int32_t)(o.mant_len - (o.dot_at < o.mant_len ? 1u : 0u));
  auto group = [&](int32_t g) -> uint32_t {
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int32_t k = 4 * g + i - o.offset;
      uint32_t d = 0;
      if (k >= 0 && k < nd) d = (uint32_t)o.mant[(uint32_t)k + ((uint32_t)k >= o.dot_at ? 1u : 0u)] - '0';
      v = v * 10u + d;
    }
    return v;
  };
  for (uint32_t j = 0; 2 * j < o.ngroups; j++) {
    const int32_t g = o.first_group + 2 * (int32_t)j;
    const uint32_t lo = group(g), hi = g + 1 <= o.last_group ? group(g + 1) : 0u;  // an odd count is zero padded to 4 bytes
    w[2 + j] = lo | (hi << 16);
  }
  return pad4(8 + 2 * o.ngroups);
}

// chrono NaiveDate::from_ymd_opt + num_days_from_ce for years 0..=9999.
DEV bool ymd_to_ce_days(uint32_t y, uint32_t m, uint32_t d, int32_t& out) {
  if (m < 1 || m > 12 || d < 1) return false;
  const bool leap = (y % 4 == 0 && y % 100 != 0) || y % 400 == 0;
  static const u8 dim[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
  uint32_t dm = dim[m - 1] + ((m == 2 && leap) ? 1 : 0);
  if (d > dm) return false;
  // days_from_civil (H. Hinnant), shifted to 0001-01-01 = day 1
  int32_t yy = (int32_t)y - (m <= 2);
  const int32_t era = (yy >= 0 ? yy : yy - 399) / 400;
  const uint32_t yoe = (uint32_t)(yy - era * 400);
  const uint32_t doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
  const uint32_t doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  out = era * 146097 + (int32_t)doe - 719468 + 719163;
  return true;
}
DEV bool two_digits(const u8* s, uint32_t& v) {
  const uint32_t h = s[0] - '0', l = s[1] - '0';
  if (h > 9 || l > 9) return false;
  v = h * 10 + l;
  return true;
}
// parse_iso_date_fast, codec/time.rs:89-100
DEV bool iso_date_fast(const u8* s, uint32_t n, int32_t& days) {
  if (n != 10 || s[4] != '-' || s[7] != '-') return false;
  uint32_t y1, y2, m, d;
  if (!two_digits(s, y1) || !two_digits(s + 2, y2) || !two_digits(s + 5, m) || !two_digits(s + 8, d)) return false;
  return ymd_to_ce_days(y1 * 100 + y2, m, d, days);
}
// parse_iso_time_fast, codec/time.rs:107-141
DEV bool iso_time_fast(const u8* s, uint32_t n, uint32_t& secs, uint32_t& nanos, bool over = false) {
  if (n < 8 || s[2] != ':' || s[5] != ':') return false;
  uint32_t h, m, sec;
  if (!two_digits(s, h) || !two_digits(s + 3, m) || !two_digits(s + 6, sec)) return false;
  nanos = 0;
  if (n != 8) {
    if (s[8] != '.') return false;
    const uint32_t fl = n - 9;
    if (fl == 0 || fl > 9) return false;
    uint32_t v = 0;
    ByteWin bw{s, over};
    for (uint32_t i = 0; i < fl; i++) { const uint32_t d = bw.at(9 + i) - '0'; if (d > 9) return false; v = v * 10 + d; }
    for (uint32_t i = fl; i < 9; i++) v *= 10;
    nanos = v;
  }
  if (h >= 24 || m >= 60 || sec >= 60) return false;
  secs = h * 3600 + m * 60 + sec;
  return true;
}
// parse_iso_timestamp_fast, codec/time.rs:145-154
DEV bool iso_timestamp_fast(const u8* s, uint32_t n, int32_t& days, uint32_t& secs, uint32_t& nanos, bool over = false) {
  if (n < 19 || s[10] != ' ') return false;
  return iso_date_fast(s, 10, days) && iso_time_fast(s + 11, n - 11, secs, nanos, over);
}
// (the register-only temporal parsers below are compiled into the kernels that ask for them — rows.hip; k_cells / k_fused keep the code their
// register allocation was tuned for: with these inlined k_cells spills 16-21 VGPRs)
#ifndef ETLG_TEMPORAL_SWAR
#define ETLG_TEMPORAL_SWAR 0
#endif
// parse_iso_timestamp_fast once more, for staged text (the loads may run past the text inside the window's slack): the 19..29 bytes
// come in as eight dwords, separators are compared in place, every two-digit field is validated and converted with the packed
// digit test of digits4, the fraction is right-padded with '0' by masks. No byte loads, no character loop. Accepts exactly what
// iso_timestamp_fast accepts and returns the same values (tests/test_gpu_parity.py::test_temporal_matrix runs both ends of every
// branch); false = "not this shape": the caller goes on with the byte-wise path.
DEV bool iso_timestamp_swar(const u8* s, uint32_t n, int32_t& days, uint32_t& secs, uint32_t& nanos) {
  if (n < 19u || n == 20u || n > 29u) return false;
  const uint32_t sh = (uint32_t)(uintptr_t)s & 3u;
  const uint32_t* q = (const uint32_t*)(s - sh);
  const uint32_t need = sh + n;
  uint32_t r[9], w[8];
#pragma unroll
  for (uint32_t j = 0; j < 9; j++) r[j] = 4 * j < need ? q[j] : 0u;
#pragma unroll
  for (uint32_t j = 0; j < 8; j++) w[j] = __builtin_amdgcn_alignbyte(r[j + 1], r[j], sh);
  const bool frac = n > 19u;
  bool sep = (w[1] & 0xFF0000FFu) == 0x2D00002Du && (w[2] & 0x00FF0000u) == 0x00200000u && (w[3] & 0x0000FF00u) == 0x00003A00u &&
             (w[4] & 0xFFu) == 0x3Au && (!frac || (w[4] >> 24) == 0x2Eu);
  if (!sep) return false;
  uint32_t badm = 0;
  auto two = [&](uint32_t x) -> uint32_t {   // two ASCII digits in bits 0..15, first digit in the low byte
    const uint32_t t = x - 0x3030u;
    badm |= ((x + 0x4646u) | t) & 0x8080u;
    return (t & 0xFu) * 10u + ((t >> 8) & 0xFu);
  };
  bool okd = true;
  const uint32_t year = digits4(w[0], okd);
  const uint32_t mon = two((w[1] >> 8) & 0xFFFFu), day = two(w[2] & 0xFFFFu);
  const uint32_t hh = two((w[2] >> 24) | ((w[3] & 0xFFu) << 8)), mi = two(w[3] >> 16), ss = two((w[4] >> 8) & 0xFFFFu);
  uint32_t nn = 0;
  if (frac) {
    const uint32_t fl = n - 20u;   // 1..9
    const uint32_t ka = fl >= 4u ? 0xFFFFFFFFu : (1u << (8u * fl)) - 1u;
    const uint32_t flb = fl > 4u ? fl - 4u : 0u;
    const uint32_t kb = flb >= 4u ? 0xFFFFFFFFu : (1u << (8u * flb)) - 1u;
    const uint32_t a = (w[5] & ka) | (0x30303030u & ~ka), b = (w[6] & kb) | (0x30303030u & ~kb);
    const uint32_t c9 = fl == 9u ? (w[7] & 0xFFu) : 0x30u;
    if (c9 - 0x30u > 9u) return false;
    nn = digits4(a, okd) * 100000u + digits4(b, okd) * 10u + (c9 - 0x30u);
  }
  if (!okd || badm || hh >= 24u || mi >= 60u || ss >= 60u) return false;
  if (!ymd_to_ce_days(year, mon, day, days)) return false;
  secs = hh * 3600u + mi * 60u + ss;
  nanos = nn;
  return true;
}
// split_timestamp_offset + parse_postgres_utc_offset for the shape Postgres prints for whole-hour zones, `...+HH` / `...-HH`, staged
// text: the index of the sign and the offset in seconds; false = another shape (the caller goes on with the general functions).
DEV bool tz_hours_swar(const u8* s, uint32_t n, uint32_t min_index, uint32_t& idx, int32_t& off) {
  if (n < min_index + 4u) return false;   // the sign must lie beyond min_index
  const uint32_t x = ldu32(s + n - 3u);
  const uint32_t sg = x & 0xFFu;
  if (sg != '+' && sg != '-') return false;
  const uint32_t d2 = (x >> 8) & 0xFFFFu, t = d2 - 0x3030u;
  if (((d2 + 0x4646u) | t) & 0x8080u) return false;
  const uint32_t h = (t & 0xFu) * 10u + ((t >> 8) & 0xFu);
  if (h >= 16u) return false;
  idx = n - 3u;
  off = sg == '-' ? -(int32_t)(h * 3600u) : (int32_t)(h * 3600u);
  return true;
}
// parse_postgres_utc_offset, crates/etl-postgres/src/time.rs:143-207
DEV bool parse_utc_offset(const u8* s, uint32_t n, int32_t& out, bool over = false) {
  if (n == 0) return false;
  int32_t sign;
  if (s[0] == '+') sign = 1; else if (s[0] == '-') sign = -1; else return false;
  s++; n--;
  uint32_t h = 0, m = 0, sec = 0;
  bool colon = false;
  { ByteWin bw{s, over}; for (uint32_t i = 0; i < n; i++) if (bw.at(i) == ':') colon = true; }
  if (colon) {
    if (n == 5) { if (s[2] != ':' || !two_digits(s, h) || !two_digits(s + 3, m)) return false; }
    else if (n == 8) { if (s[2] != ':' || s[5] != ':' || !two_digits(s, h) || !two_digits(s + 3, m) || !two_digits(s + 6, sec)) return false; }
    else return false;
  } else {
    if (n == 2) { if (!two_digits(s, h)) return false; }
    else if (n == 4) { if (!two_digits(s, h) || !two_digits(s + 2, m)) return false; }
    else if (n == 6) { if (!two_digits(s, h) || !two_digits(s + 2, m) || !two_digits(s + 4, sec)) return false; }
    else return false;
  }
  if (m >= 60 || sec >= 60) return false;
  const uint32_t total = h * 3600 + m * 60 + sec;
  if (total >= 16 * 3600) return false;
  out = sign * (int32_t)total;
  return true;
}
// split_utc_offset / split_timestamp_offset: last '+'/'-' with byte index > min_index.
DEV int32_t split_offset_index(const u8* s, uint32_t n, uint32_t min_index, bool over = false) {
  ByteWin bw{s, over};
  for (uint32_t i = n; i-- > 0;) {
    if (i <= min_index) break;
    const uint32_t c = bw.at_rev(i);
    if (c == '+' || c == '-') return (int32_t)i;
  }
  return -1;
}
// ---- The temporal shapes outside the fixed-layout fast paths: what the reference hands to chrono (codec/time.rs:21-55: "anything
// else falls back to chrono's format machinery"). chrono 0.4.44 (Cargo.lock:1176) is not vendored under /root/reference; its
// `parse_from_str` is restated here from its published behaviour for the three format strings the reference uses
// (crates/etl-postgres/src/time.rs:13-21: "%Y-%m-%d", "%H:%M:%S%.f", "%Y-%m-%d %H:%M:%S%.f"): a numeric item first skips
// whitespace (char::is_whitespace), then takes 1..=2 digits (%m %d %H %M %S) or 1..=4 (%Y; any count behind an explicit '+' / '-');
// literals match exactly; the format's space matches zero or more whitespace; "%.f" is nothing, or '.' and at least one digit (nine
// are significant, more are skipped); input left over is an error. Resolution: NaiveDate::from_ymd_opt over years
// -262143..=262142, hour < 24, minute < 60, second <= 60 with 60 a leap second (second 59, nanos + 10^9). KATs: codec/time.rs:181-269.
// Rare path (Postgres emits the ISO shapes): out of line, byte loads.
#ifndef ETLG_CHRONO_CALL
#define ETLG_CHRONO_CALL static __device__ __attribute__((noinline))
#endif
constexpr int32_t kChronoMinDays = -95746129, kChronoMaxDays = 95745399;   // num_days_from_ce of -262143-01-01 and 262142-12-31
struct ChronoCur {
  const u8* s; uint32_t n, i;
  DEV void skip_ws() { for (;;) { const uint32_t l = ws_len_at(s + i, n - i); if (!l) break; i += l; } }
  DEV bool lit(uint32_t c) { if (i < n && s[i] == c) { i++; return true; } return false; }
  // 1..=maxw digits; the value saturates (a year that large fails the range check, as chrono's i64 overflow fails the parse)
  DEV bool digits(uint32_t maxw, uint32_t& v) {
    uint32_t k = 0; v = 0;
    for (; i < n && k < maxw; i++, k++) { const uint32_t d = (uint32_t)s[i] - '0'; if (d > 9) break; v = v > 99999999u ? v : v * 10u + d; }
    return k > 0;
  }
  DEV bool num2(uint32_t& v) { skip_ws(); return digits(2, v); }
  DEV bool date(int32_t& days) {
    skip_ws();
    uint32_t y, m, d; bool neg = false, ok;
    if (i < n && (s[i] == '-' || s[i] == '+')) { neg = s[i] == '-'; i++; ok = digits(0xFFFFFFFFu, y); } else ok = digits(4, y);
    if (!ok || !lit('-') || !num2(m) || !lit('-') || !num2(d)) return false;
    const int32_t yy = neg ? -(int32_t)y : (int32_t)y;
    if (yy < -262143 || yy > 262142 || m < 1 || m > 12 || d < 1) return false;
    const bool leap = (yy % 4 == 0 && yy % 100 != 0) || yy % 400 == 0;
    const uint32_t dm = m == 2 ? (leap ? 29u : 28u) : (m == 4 || m == 6 || m == 9 || m == 11) ? 30u : 31u;
    if (d > dm) return false;
    const int32_t y2 = yy - (m <= 2 ? 1 : 0);
    const int32_t era = (y2 >= 0 ? y2 : y2 - 399) / 400;
    const uint32_t yoe = (uint32_t)(y2 - era * 400);
    const uint32_t doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
    days = era * 146097 + (int32_t)(yoe * 365 + yoe / 4 - yoe / 100 + doy) - 719468 + 719163;
    return true;
  }
  DEV bool time(uint32_t& secs, uint32_t& nanos) {
    uint32_t h, m, sec;
    if (!num2(h) || !lit(':') || !num2(m) || !lit(':') || !num2(sec)) return false;
    nanos = 0;
    if (i < n && s[i] == '.') {
      i++;
      uint32_t k = 0;
      for (; i < n; i++, k++) { const uint32_t d = (uint32_t)s[i] - '0'; if (d > 9) break; if (k < 9) nanos = nanos * 10u + d; }
      if (!k) return false;
      for (; k < 9; k++) nanos *= 10u;
    }
    if (h >= 24 || m >= 60 || sec > 60) return false;
    if (sec == 60) { sec = 59; nanos += 1000000000u; }
    secs = h * 3600 + m * 60 + sec;
    return true;
  }
};
// The whole of parse_postgres_{date,time,timestamp} / the timestamp and time halves of timestamptz / timetz for texts the fast
// paths turned down. `n`: the text without its offset part (timestamptz / timetz: the caller split it off); trailing whitespace
// is trimmed for those two (`timestamp.trim_end()`, codec/time.rs:67; `time.trim_end()`, etl-postgres/src/time.rs:123).
// out: days (date classes), secs, nanos. Returns false for chrono's ParseError.
ETLG_CHRONO_CALL bool chrono_fallback(uint32_t cls, const u8* s, uint32_t n, uint32_t* out3) {
  if (cls == ETLG_TC_TIMESTAMPTZ || cls == ETLG_TC_TIMETZ) {
    for (;;) {   // str::trim_end
      if (!n) break;
      uint32_t j = n - 1;
      while (j > 0 && (s[j] & 0xC0) == 0x80) j--;
      const uint32_t l = ws_len_at(s + j, n - j);
      if (!l || l != n - j) break;
      n = j;
    }
  }
  ChronoCur c{s, n, 0};
  int32_t days = 0; uint32_t secs = 0, nanos = 0;
  bool ok = true;
  if (cls == ETLG_TC_DATE || cls == ETLG_TC_TIMESTAMP || cls == ETLG_TC_TIMESTAMPTZ) ok = c.date(days);
  if (ok && (cls == ETLG_TC_TIMESTAMP || cls == ETLG_TC_TIMESTAMPTZ)) c.skip_ws();
  if (ok && cls != ETLG_TC_DATE) ok = c.time(secs, nanos);
  if (!ok || c.i != c.n) return false;
  out3[0] = (uint32_t)days; out3[1] = secs; out3[2] = nanos;
  return true;
}

DEV int hexv(uint32_t c) {
  if (c - '0' < 10u) return (int)(c - '0');
  c = lower(c);
  if (c - 'a' < 6u) return (int)(c - 'a' + 10);
  return -1;
}
// uuid 1.23 Uuid::parse_str: simple(32) | hyphenated(36) | {braced}(38) | urn:uuid:(45)
// Four ASCII hex characters per 32-bit word (character i in byte i): validity of all four at once, then their two bytes.
DEV bool hex4_ok(uint32_t x) {
  const uint32_t x7 = x & 0x7F7F7F7Fu;
  const uint32_t t = x7 ^ 0x30303030u;                       // '0'..'9' -> 0..9
  const uint32_t dig = ~(t + 0x76767676u);                   // bit 7 of a byte set <=> t < 10
  const uint32_t a = (x7 | 0x20202020u) ^ 0x60606060u;       // 'A'..'F', 'a'..'f' -> 1..6
  const uint32_t alp = (a + 0x7F7F7F7Fu) & ~(a + 0x79797979u);  // bit 7 set <=> 1 <= a <= 6
  return (((dig | alp) & ~x) & 0x80808080u) == 0x80808080u;  // ... and the character itself is ASCII
}
DEV uint32_t hex4_bytes(uint32_t x) {  // valid input only: 2 bytes, first character pair in bits 0..7
  const uint32_t n = (x & 0x0F0F0F0Fu) + 9u * ((x >> 6) & 0x01010101u);
  const uint32_t p = (n << 4) | (n >> 8);
  return (p & 0xFFu) | ((p >> 8) & 0xFF00u);
}
DEV bool parse_uuid(const u8* s, uint32_t n, uint32_t* out4, bool over = false) {
  uint32_t skip;  // bytes in front of the 32 / 36 characters
  if (n == 32 || n == 36) skip = 0;
  else if (n == 38) { if (s[0] != '{' || s[37] != '}') return false; skip = 1; }
  else if (n == 45) {
    if (!(s[0] == 'u' && s[1] == 'r' && s[2] == 'n' && s[3] == ':' && s[4] == 'u' && s[5] == 'u' && s[6] == 'i' && s[7] == 'd' && s[8] == ':')) return false;
    skip = 9;
  } else return false;
  const u8* h = s + skip;
  const uint32_t sh = (uint32_t)(uintptr_t)h & 3u;
  const uint32_t* q = (const uint32_t*)(h - sh);
  // characters [4 i, 4 i + 4) of the text. Staged bytes: aligned dwords + v_alignbyte (reads end inside the dword after the last character)
  auto word = [&](uint32_t i) -> uint32_t {
    if (over) return __builtin_amdgcn_alignbyte(q[i + 1], q[i], sh);
    return (uint32_t)h[4 * i] | ((uint32_t)h[4 * i + 1] << 8) | ((uint32_t)h[4 * i + 2] << 16) | ((uint32_t)h[4 * i + 3] << 24);
  };
  // two halves of sixteen digits each, so that at most five words of text are live at a time (all nine at once cost k_cells
  // its fourth workgroup per CU without spills)
  bool ok = true;
  if (n == 32) {
    {
      const uint32_t a = word(0), b = word(1), c = word(2), d = word(3);
      ok = hex4_ok(a) & hex4_ok(b) & hex4_ok(c) & hex4_ok(d);
      out4[0] = hex4_bytes(a) | (hex4_bytes(b) << 16); out4[1] = hex4_bytes(c) | (hex4_bytes(d) << 16);
    }
    {
      const uint32_t a = word(4), b = word(5), c = word(6), d = word(7);
      ok &= hex4_ok(a) & hex4_ok(b) & hex4_ok(c) & hex4_ok(d);
      out4[2] = hex4_bytes(a) | (hex4_bytes(b) << 16); out4[3] = hex4_bytes(c) | (hex4_bytes(d) << 16);
    }
    return ok;
  }
  {  // 8-4-4-4-12: "xxxxxxxx-xxxx-xxxx-" are characters 0..18
    const uint32_t a = word(0), b = word(1), w2 = word(2), w3 = word(3), w4 = word(4);
    ok = (w2 & 0xFFu) == '-' && ((w3 >> 8) & 0xFFu) == '-' && ((w4 >> 16) & 0xFFu) == '-';
    const uint32_t c = __builtin_amdgcn_alignbyte(w3, w2, 1), d = __builtin_amdgcn_alignbyte(w4, w3, 2);
    ok &= hex4_ok(a) & hex4_ok(b) & hex4_ok(c) & hex4_ok(d);
    out4[0] = hex4_bytes(a) | (hex4_bytes(b) << 16); out4[1] = hex4_bytes(c) | (hex4_bytes(d) << 16);
  }
  {  // "xxxx-xxxxxxxxxxxx" are characters 19..35
    const uint32_t w4 = word(4), w5 = word(5), b = word(6), c = word(7), d = word(8);
    ok &= (w5 >> 24) == '-';
    const uint32_t a = __builtin_amdgcn_alignbyte(w5, w4, 3);
    ok &= hex4_ok(a) & hex4_ok(b) & hex4_ok(c) & hex4_ok(d);
    out4[2] = hex4_bytes(a) | (hex4_bytes(b) << 16); out4[3] = hex4_bytes(c) | (hex4_bytes(d) << 16);
  }
  return ok;
}

// f32 / f64 `str::parse`: the exact fast path lives in float_fast.h (host-testable); here it reads through ByteWin.
#ifndef ETLG_FLOAT_CALL   // k_cells calls it out of line: inlined, its two call sites (sizing, decode) cost the kernel ~25 VGPRs, i.e. a workgroup per CU
#define ETLG_FLOAT_CALL DEV
#endif
ETLG_FLOAT_CALL int parse_float_fast(const u8* s, uint32_t n, bool is32, uint64_t& out, bool over) {
  if (over && n <= 24) {  // up to 24 bytes: three loads, then register arithmetic without a per-character loop (float_fast.h, parse_float_swar)
    const uint64_t x0 = ldu64(s), x1 = n > 8 ? ldu64(s + 8) : 0ull, x2 = n > 16 ? ldu64(s + 16) : 0ull;
    const int r = parse_float_swar(x0, x1, x2, n, is32, out);
    if (r != 3) return r;
  }
  ByteWin bw{s, over};
  return parse_float_fast_t([&](uint32_t i) { return bw.at(i); }, n, is32, out);
}

// Which classes are handed back DEFERRED wholesale (include/etlg.h contract).
DEV bool class_always_deferred(uint32_t cls) {
  return cls == ETLG_TC_JSON || cls == ETLG_TC_ARRAY;
}

// Heap bytes a text cell will occupy (exact for every non-error outcome).
DEV uint32_t cell_heap_bytes(uint32_t cls, const u8* d, uint32_t len, bool over = false) {
  if (class_always_deferred(cls) || cls == ETLG_TC_STRING) return pad4(len);
  switch (cls) {
    case ETLG_TC_BYTEA: return len >= 2 ? pad4((len - 2) >> 1) : 0;
    case ETLG_TC_F32: case ETLG_TC_F64: { uint64_t b; return parse_float_fast(d, len, cls == ETLG_TC_F32, b, over) == 1 ? pad4(len) : 0; }
    case ETLG_TC_NUMERIC: { NumShape s; return ((over && numeric_plain(d, len, s)) || numeric_scan(d, len, s, over)) ? pad4(8 + 2 * s.ngroups) : 0; }
    // date / time / timestamp / timestamptz / timetz: decoded on the device whatever their shape (chrono_fallback): no heap entry
    default: return 0;
  }
}

// Copies n bytes to a 4-byte aligned heap position, zero padded to 4.
DEV void heap_copy(u8* dst, const u8* src, uint32_t n) {
  uint32_t* w = (uint32_t*)dst;
  uint32_t i = 0;
  for (; i + 4 <= n; i += 4) { uint32_t v; __builtin_memcpy(&v, src + i, 4); w[i >> 2] = v; }
  if (i < n) {
    uint32_t v = 0;
    for (uint32_t k = 0; i + k < n; k++) v |= (uint32_t)src[i + k] << (8 * k);
    w[i >> 2] = v;
  }
}

// parse_cell_from_postgres_text for one cell (codec/text.rs:32-153), writing the
// slot words / heap entry. Returns 0 or an etlg_err_code; `state` = cell state.
DEV uint32_t slot_bytes(uint32_t cls);
// COPY_CLASSES = false: the caller moves verbatim text (String cells, wholesale-deferred classes)
// itself and never passes those classes in (k_cells).
template <bool COPY_CLASSES = true>
DEV_DECODE uint32_t decode_text_cell(uint32_t cls, const u8* d, uint32_t len, uint32_t* slot, u8* heap, uint32_t& hcur,
                              uint32_t& state, bool over = false) {
  // str::from_utf8 precedes the type switch (codec/event.rs:976). Every non-text grammar below
  // accepts ASCII only, so for those classes validity is only examined when the parse fails
  // (to pick between the UTF-8 error and the type's own error) or before a cell is deferred.
  state = ETLG_CELL_VALUE;
  auto bad = [&](uint32_t code) { return utf8_valid(d, len) ? code : (uint32_t)ETLG_E_UTF8; };
  auto var = [&](uint32_t nbytes) { slot[0] = hcur; slot[1] = nbytes; hcur += pad4(nbytes); };
  auto defer = [&]() {
    if (!utf8_valid(d, len)) return (uint32_t)ETLG_E_UTF8;
    state = ETLG_CELL_DEFERRED; heap_copy(heap + hcur, d, len); var(len);
    if (slot_bytes(cls) == 12) slot[2] = 0;
    return 0u;
  };
  if (!COPY_CLASSES && (cls == ETLG_TC_STRING || class_always_deferred(cls))) return 0;
  if (class_always_deferred(cls)) return defer();
  switch (cls) {
    case ETLG_TC_STRING:
      if (!utf8_valid(d, len)) return ETLG_E_UTF8;
      heap_copy(heap + hcur, d, len); var(len); return 0;
    case ETLG_TC_BOOL:  // parse_bool, codec/bool.rs:11-19
      if (len == 1 && (d[0] == 't' || d[0] == 'f')) { slot[0] = d[0] == 't'; return 0; }
      return bad(ETLG_E_BOOL);
    case ETLG_TC_I16: case ETLG_TC_I32: case ETLG_TC_U32: case ETLG_TC_I64: {  // one copy of the integer parser for all widths
      const bool sg = cls != ETLG_TC_U32;
      const int bits = cls == ETLG_TC_I16 ? 16 : cls == ETLG_TC_I64 ? 64 : 32;
      int64_t v;
      bool int_ok;
      if (over) int_ok = parse_int_swar(d, len, sg, bits, v);
      else { int64_t slow = 0; int_ok = parse_int(d, len, sg, bits, slow); v = slow; }  // `v` never has its address taken by an out-of-line call
      if (!int_ok) return bad(ETLG_E_INT);
      if (cls == ETLG_TC_I64) st64(slot, (uint64_t)v); else slot[0] = (uint32_t)v;
      return 0;
    }
    case ETLG_TC_F32: case ETLG_TC_F64: {
      uint64_t b;
      const int r = parse_float_fast(d, len, cls == ETLG_TC_F32, b, over);
      if (r == 2) return bad(ETLG_E_FLOAT);
      if (r == 1) return defer();
      st64(slot, b);
      return 0;
    }
    case ETLG_TC_NUMERIC: {
      NumShape s;
      if (over && numeric_plain(d, len, s)) {  // digits, one '.', a sign: ASCII by construction
        numeric_emit_plain(s, heap + hcur);
        var(8 + 2 * s.ngroups);
        return 0;
      }
      if (!numeric_scan(d, len, s, over)) return bad(ETLG_E_NUMERIC);
      // the numeric grammar strips Unicode whitespace, so a successful scan may still
      // have seen multi-byte characters: those must be valid UTF-8 too
      if (!utf8_valid(d, len)) return ETLG_E_UTF8;
      numeric_emit(s, heap + hcur, over);
      var(8 + 2 * s.ngroups);
      return 0;
    }
    case ETLG_TC_BYTEA: {  // parse_bytea_hex_string, codec/hex.rs:11-52
      if (len < 2 || d[0] != '\\' || d[1] != 'x') return bad(ETLG_E_BYTEA);
      if ((len - 2) & 1) return bad(ETLG_E_BYTEA);
      const uint32_t nb = (len - 2) >> 1;
      u8* dst = heap + hcur;
      uint32_t w = 0;
      ByteWin bw{d, over};
      for (uint32_t i = 0; i < nb; i++) {
        int h = hexv(bw.at(2 + 2 * i));
        if (h < 0) return bad(ETLG_E_BYTEA);
        int l = hexv(bw.at(3 + 2 * i));
        if (l < 0) return bad(ETLG_E_BYTEA);
        w |= (uint32_t)((h << 4) | l) << (8 * (i & 3));
        if ((i & 3) == 3) { ((uint32_t*)dst)[i >> 2] = w; w = 0; }
      }
      if (nb & 3) ((uint32_t*)dst)[nb >> 2] = w;
      var(nb);
      return 0;
    }
    case ETLG_TC_DATE: {
      int32_t x;
      if (iso_date_fast(d, len, x)) { slot[0] = (uint32_t)x; slot[1] = 0; return 0; }
      uint32_t o[3];
      if (!chrono_fallback(cls, d, len, o)) return bad(ETLG_E_DATETIME);
      slot[0] = o[0]; slot[1] = 0; return 0;
    }
    case ETLG_TC_TIME: {
      uint32_t a, b;
      if (iso_time_fast(d, len, a, b, over)) { slot[0] = a; slot[1] = b; return 0; }
      uint32_t o[3];
      if (!chrono_fallback(cls, d, len, o)) return bad(ETLG_E_DATETIME);
      slot[0] = o[1]; slot[1] = o[2]; return 0;
    }
    case ETLG_TC_TIMESTAMP: {
      int32_t x; uint32_t a, b;
      if ((ETLG_TEMPORAL_SWAR && over && iso_timestamp_swar(d, len, x, a, b)) || iso_timestamp_fast(d, len, x, a, b, over)) { slot[0] = (uint32_t)x; slot[1] = a; slot[2] = b; return 0; }
      uint32_t o[3];
      if (!chrono_fallback(cls, d, len, o)) return bad(ETLG_E_DATETIME);
      slot[0] = o[0]; slot[1] = o[1]; slot[2] = o[2];
      return 0;
    }
    case ETLG_TC_TIMESTAMPTZ: {  // codec/time.rs:63-71 + UTC normalisation codec/text.rs:108-111
      int32_t x; uint32_t a, b;
      int32_t off;
      uint32_t fidx;
      if (ETLG_TEMPORAL_SWAR && over && tz_hours_swar(d, len, 10, fidx, off) && iso_timestamp_swar(d, fidx, x, a, b)) {
        // (the shape Postgres prints: both halves decided without a byte load)
      } else {
      const int32_t idx = split_offset_index(d, len, 10, over);
      if (idx < 0) return bad(ETLG_E_DATETIME);
      if (!iso_timestamp_fast(d, (uint32_t)idx, x, a, b, over)) {
        uint32_t o[3];
        if (!chrono_fallback(cls, d, (uint32_t)idx, o)) return bad(ETLG_E_DATETIME);
        x = (int32_t)o[0]; a = o[1]; b = o[2];
      }
      if (!parse_utc_offset(d + idx, len - (uint32_t)idx, off, over)) return bad(ETLG_E_DATETIME);
      }
      int32_t sec = (int32_t)a - off;
      if (sec < 0) { sec += 86400; x -= 1; } else if (sec >= 86400) { sec -= 86400; x += 1; }
      if (x < kChronoMinDays || x > kChronoMaxDays) return bad(ETLG_E_DATETIME);   // from_local_datetime(..).single() is None outside NaiveDate's range
      slot[0] = (uint32_t)x; slot[1] = (uint32_t)sec; slot[2] = b;
      return 0;
    }
    case ETLG_TC_TIMETZ: {  // crates/etl-postgres/src/time.rs:121-127 (NaiveTime::parse_from_str on the trimmed time part)
      const int32_t idx = split_offset_index(d, len, 0, over);
      if (idx < 0) return bad(ETLG_E_DATETIME);
      uint32_t a, b;
      if (!iso_time_fast(d, (uint32_t)idx, a, b, over)) {
        uint32_t o[3];
        if (!chrono_fallback(cls, d, (uint32_t)idx, o)) return bad(ETLG_E_DATETIME);
        a = o[1]; b = o[2];
      }
      int32_t off;
      if (!parse_utc_offset(d + idx, len - (uint32_t)idx, off, over)) return bad(ETLG_E_DATETIME);
      slot[0] = a; slot[1] = b; slot[2] = (uint32_t)off;
      return 0;
    }
    case ETLG_TC_UUID: return parse_uuid(d, len, slot, over) ? 0 : bad(ETLG_E_UUID);
    default: return defer();
  }
}

DEV uint32_t slot_bytes(uint32_t cls) {  // layout rule of include/etlg.h
  switch (cls) {
    case ETLG_TC_BOOL: case ETLG_TC_I16: case ETLG_TC_I32: case ETLG_TC_U32: return 4;
    case ETLG_TC_TIMESTAMP: case ETLG_TC_TIMESTAMPTZ: case ETLG_TC_TIMETZ: return 12;
    case ETLG_TC_UUID: return 16;
    default: return 8;
  }
}

// ---------------------------------------------------------------- k_size
// Heap bytes of one tuple decoded against `n` schema columns.
//   mode 0: full row (convert_tuple_to_row); mode 1: dense key tuple;
//   mode 2: full-width key tuple (non-identity positions skipped unread);
//   mode 3: update new tuple (u cells never allocate: alias or Missing)
DEV uint32_t tuple_heap_bytes(const DecParams& p, const DevSlot& s, const u8* tuple, uint32_t ncells, int mode, bool over = false) {
  if (!s.has_var) return 0;  // fixed-width schema: no cell can reach the heap
  const DevCol* cols = p.cols + s.cols_base;
  CellIt it; it.begin(tuple);
  uint32_t h = 0, next_ident = 0;
  for (uint32_t i = 0; i < ncells; i++) {
    const u8* d; uint32_t len;
    const uint32_t t = it.next(d, len);
    uint32_t ci;
    if (mode == 1) {
      // i-th identity column
      ci = 0xFFFFFFFFu;
      for (uint32_t c = 0, k = 0; c < s.n_cols; c++) if (cols[c].identity) { if (k == i) { ci = c; break; } k++; }
      if (ci == 0xFFFFFFFFu) break;
    } else {
      if (i >= s.n_cols) break;
      ci = i;
      if (mode == 2) { if (!cols[ci].identity) continue; next_ident++; }
    }
    if (t == 't') h += cell_heap_bytes(cols[ci].cls, d, len, over);
  }
  (void)next_ident;
  return h;
}

// ---------------------------------------------------------------- k_write
// Per-cell states are 2 bits, 16 columns per 32-bit word at the head of the row;
// they are accumulated in a register and stored once per word. Every slot word is
// written exactly once (zeros for NULL / MISSING cells), so rows need no pre-clear.
struct StateAcc {
  uint32_t* words; uint32_t acc = 0;
  DEV void put(uint32_t i, uint32_t st, bool last) {
    acc |= st << (2 * (i & 15));
    if ((i & 15) == 15 || last) { words[i >> 4] = acc; acc = 0; }
  }
};
DEV void slot_zero(uint32_t* slot, uint32_t cls) {
  const uint32_t nw = slot_bytes(cls) >> 2;
  for (uint32_t w = 0; w < nw; w++) slot[w] = 0;
}
DEV uint32_t get_state(const u8* row, uint32_t i) { return (((const uint32_t*)row)[i >> 4] >> (2 * (i & 15))) & 3u; }

// One row image out of one tuple. `mode`:
//   ROW_FULL    convert_tuple_to_row (codec/event.rs:554-587), full layout
//   ROW_KEY     normalize_key_tuple_to_row (codec/event.rs:795-923), key layout (dense or full-width tuple)
//   ROW_UPDATE  convert_update_tuple_to_updated_table_row + OldRowResolver (codec/event.rs:605-791):
//               'u' cells alias the aligned old value (Cell::clone) or become MISSING (Partial)
// A single body (one inlined copy of the value codec) serves all three.
enum : uint32_t { ROW_FULL = 0, ROW_KEY = 1, ROW_UPDATE = 2 };

DEV uint32_t write_row(const DecParams& p, const DevSlot& s, uint32_t mode, const u8* tuple, uint32_t ncells, u8* row,
                       uint32_t old_kind, const u8* old_row, uint32_t& hcur, bool& partial, bool over = false) {
  bool dense = false;
  if (mode == ROW_KEY) {
    if (s.n_ident == 0) return ETLG_E_KEY_MISSING_COLS;
    dense = ncells == s.n_ident;
    if (!dense && ncells != s.n_cols) return ETLG_E_KEY_SHAPE;
  } else if (ncells != s.n_cols) {
    return ETLG_E_TUPLE_WIDTH;
  }
  const DevCol* cols = p.cols + s.cols_base;
  StateAcc sa{(uint32_t*)row};
  CellIt it; it.begin(tuple);
  uint32_t ci = 0;  // schema column cursor (dense key tuples walk identity columns only)
  const uint32_t n_out = mode == ROW_KEY ? s.n_ident : s.n_cols;
  // A decoded old row always has exactly n_cols (Full) / n_ident (Key) cells, so the
  // resolver's width checks (codec/event.rs:700-710, 730-740, 772-785) cannot fire here.
  for (uint32_t i = 0; i < ncells; i++) {
    const u8* d; uint32_t len;
    const uint32_t t = it.next(d, len);
    if (mode == ROW_KEY) {
      if (dense) { while (ci < s.n_cols && !cols[ci].identity) ci++; }
      else { ci = i; if (!cols[ci].identity) continue; }  // full-width key tuple: other positions are skipped unread
    } else {
      ci = i;
    }
    const DevCol col = cols[ci];
    const uint32_t k = mode == ROW_KEY ? col.key_index : i;  // output cell index
    uint32_t* slot = (uint32_t*)(row + (mode == ROW_KEY ? col.off_key : col.off_full));
    uint32_t st = ETLG_CELL_NULL;
    if (t == 'n') {  // convert_tuple_data_to_cell, codec/event.rs:945-961
      if (!col.nullable && !(p.flags & 2u)) return ETLG_E_REQUIRED_NULL;
      slot_zero(slot, col.cls);
    } else if (t == 'u') {
      if (mode == ROW_FULL) return ETLG_E_FULL_ROW_MISSING;
      if (mode == ROW_KEY) return ETLG_E_KEY_MISSING_VALUE;
      const bool from_full = old_kind == ETLG_OLD_FULL;
      const bool from_key = old_kind == ETLG_OLD_KEY && col.identity;
      if (from_full || from_key) {  // Cell::clone of the aligned old value: alias its slot
        const uint32_t oi = from_full ? i : col.key_index;
        const uint32_t* src = (const uint32_t*)(old_row + (from_full ? col.off_full : col.off_key));
        const uint32_t nw = slot_bytes(col.cls) >> 2;
        for (uint32_t w = 0; w < nw; w++) slot[w] = src[w];
        st = get_state(old_row, oi);
      } else {
        slot_zero(slot, col.cls);
        st = ETLG_CELL_MISSING;
        partial = true;
      }
    } else if (t == 't') {
      const uint32_t err = decode_text_cell(col.cls, d, len, slot, p.heap, hcur, st, over);
      if (err) return err;
    } else {
      return ETLG_E_BINARY_FORMAT;
    }
    sa.put(k, st, k + 1 == n_out);
    if (mode == ROW_KEY && dense) ci++;
  }
  return 0;
}

// convert_tuple_to_row for a wave whose active lanes ALL decode an INSERT of the same
// schema slot with the same column count (the common case). `slot_u` / `n` are wave-uniform
// (SGPRs): the column descriptors come through the scalar cache, the loop trip count and
// the per-column class dispatch are scalar branches, and only the data-dependent work
// (cell tag, length, characters) stays per lane. `pg` must hold GLOBAL side-table pointers.
DEV uint32_t write_full_row_uniform(const DecParams& pg, uint32_t slot_u, const u8* tuple, uint32_t n, u8* row,
                                    uint32_t& hcur, bool over) {
  // The descriptors are read as whole dwords through constant-address-space pointers: the address is wave-uniform
  // and the memory is never written by a kernel, so these are s_load (scalar cache, results in SGPRs). As byte /
  // halfword fields of a plain global struct they compiled to global_load_ubyte / _ushort + s_waitcnt vmcnt(0) +
  // v_readfirstlane — one dependent vector-memory round trip per column (gfx950 has no sub-dword scalar loads).
  const ETLG_CONST_AS uint32_t* sw = (const ETLG_CONST_AS uint32_t*)(uintptr_t)(pg.slots + slot_u);
  static_assert(sizeof(DevSlot) == 44 && sizeof(DevCol) == 12, "descriptor words below");
  if (n != sw[0]) return ETLG_E_TUPLE_WIDTH;                      // DevSlot.n_cols
  const ETLG_CONST_AS uint32_t* cw = (const ETLG_CONST_AS uint32_t*)(uintptr_t)(pg.cols + sw[6]);  // DevSlot.cols_base
  struct { uint32_t cls, nullable, off_full; } col;
  const u8* c = tuple + 2;
  uint32_t acc = 0;
  uint32_t* stw = (uint32_t*)row;
  for (uint32_t i = 0; i < n; i++) {
    { const uint32_t w0 = cw[3 * i], w1 = cw[3 * i + 1]; col.cls = w0 & 0xFFu; col.nullable = (w0 >> 8) & 0xFFu; col.off_full = w1 & 0xFFFFu; }
    const uint32_t cls = col.cls;
    uint32_t* slot = (uint32_t*)(row + col.off_full);  // always the arena: a pointer that may also name private memory turns every row store into a flat store
    uint64_t head = 0;
    if (over) __builtin_memcpy(&head, c, 8);  // tag + length in one load
    const uint32_t t = over ? (uint32_t)head & 0xFFu : (uint32_t)*c;
    c++;
    uint32_t st = ETLG_CELL_NULL;
    if (t == 't') {
      const uint32_t len = over ? __builtin_bswap32((uint32_t)(head >> 8)) : ld_be32(c);
      c += 4;
      const uint32_t err = decode_text_cell(cls, c, len, slot, pg.heap, hcur, st, over);
      if (err) return err;
      c += len;
    } else if (t == 'n') {
      if (!col.nullable && !(pg.flags & 2u)) return ETLG_E_REQUIRED_NULL;
      slot_zero(slot, cls);
    } else if (t == 'u') {
      return ETLG_E_FULL_ROW_MISSING;
    } else {
      return ETLG_E_BINARY_FORMAT;
    }
    acc |= st << (2 * (i & 15));
    if (((i & 15) == 15 || i + 1 == n) && !(pg.flags & 0x100u)) { stw[i >> 4] = acc; acc = 0; }
  }
  return 0;
}

// ------------------------------------------------------------ per-frame passes
// Transaction context of one frame (state BEFORE the frame is applied).
struct TxnCtx {
  bool in_txn;         // remote_final_lsn.is_some()
  uint64_t final_lsn;  // valid when in_txn
  uint64_t ord;        // ordinal this frame takes if it consumes one
};

struct FrameView {
  uint32_t f;     // frame index in the batch
  uint32_t tag;   // pgoutput tag ('k' keepalive, FT_BAD malformed)
  const u8* fr;   // first byte of the CopyData frame ('d')
  const u8* e;    // one past its last byte
};

// Structural validation of the whole message (what the third-party parser does
// before the reference sees it). For I/U/D fills `m`. Returns false on a wire error.
DEV bool frame_structure(const FrameView& v, RowMsg& m, bool over = false) {
  const u8* b = v.fr + kBodyOff;
  const u8* e = v.e;
  switch (v.tag) {
    case FT_BAD: return false;
    case 'O': { const u8* c = b + 8; return e - b >= 9 && has_cstr(c, e); }              // u64 lsn, cstr name
    case 'Y': { const u8* c = b + 4; return e - b >= 6 && has_cstr(c, e) && has_cstr(c, e); }  // u32 oid, cstr, cstr
    case 'R': return e - b >= 4;  // the rest of R is validated by the host control plane
    case 'I': case 'U': case 'D': return parse_row_msg(v.tag, b, e, m, over);
    case 'T': {  // i32 nrel, i8 options, nrel x u32
      if (e - b < 5) return false;
      const uint32_t nrel = ld_be32(b);
      return !(nrel & 0x80000000u) && (uint64_t)(e - b - 5) / 4 >= nrel;
    }
    default: return true;  // 'k', 'B', 'C' were length-checked by classify; 'M' is the host's
  }
}

// Transaction state machine + ownership + exact output sizes of one frame
// (apply.rs:2279-2617, 2836-2867). `wire_ok`/`m` come from frame_structure.
// check_txn = false: the transaction context is not known yet (fused kernel, early
// sizing); the frame is sized as if inside a transaction and txn_check_frame runs later.
DEV void size_frame(const DecParams& p, const FrameView& v, const TxnCtx& tx, bool wire_ok, const RowMsg& m,
                    uint32_t& emit, uint32_t& fixed, uint32_t& heap, uint64_t pay[3], int& row_slot,
                    bool check_txn = true, bool over = false) {
  const uint32_t f = v.f, tag = v.tag;
  const u8* b = v.fr + kBodyOff;
  emit = 0; fixed = 0; heap = 0;
  if (!wire_ok) { record_error(p, f, RK_WIRE, ETLG_E_WIRE); return; }
  switch (tag) {
    case 'B': emit = 1; fixed = 8; break;
    case 'C':
      if (check_txn && !tx.in_txn) record_error(p, f, RK_TXN, ETLG_E_TXN_STATE);
      else if (check_txn && ld_be64(b + 1) != tx.final_lsn) record_error(p, f, RK_TXN, ETLG_E_COMMIT_LSN);
      else { emit = 1; fixed = 16; }
      break;
    case 'M':
      if (p.flags & 1u) record_error(p, f, RK_WIRE, ETLG_E_CTRL_HINT);
      break;  // host control plane (apply.rs:2160-2276)
    case 'R': {  // handle_relation_message: the event is emitted here, the schema work is the host's
      if (p.flags & 1u) { record_error(p, f, RK_WIRE, ETLG_E_CTRL_HINT); break; }
      if (check_txn && !tx.in_txn) { record_error(p, f, RK_TXN, ETLG_E_TXN_STATE); break; }
      const uint32_t rel = ld_be32(b);
      const DevEpoch* ep = epoch_at(p, find_table(p, rel), f);
      if (ep && ep->emit) emit = 1;
      break;
    }
    case 'I': case 'U': case 'D': {
      if (check_txn && !tx.in_txn) { record_error(p, f, RK_TXN, ETLG_E_TXN_STATE); break; }
      pay[tag == 'I' ? 0 : tag == 'U' ? 1 : 2] = m.vbytes;  // metrics precede the ownership check
      int slot;
      if (p.flags & 2u) slot = p.copy_slot;  // table-copy rows: the caller named the schema
      else {
        const int ti = find_table(p, m.rel_id);
        if (!should_apply(p, ti, m.rel_id, tx.final_lsn)) break;
        slot = cache_slot_before(p, ti, f);
        if (slot < 0) { record_error(p, f, RK_SCHEMA, (uint32_t)(-slot)); break; }
      }
      const DevSlot& s = p.slots[slot];
      row_slot = slot;
      emit = 1;
      for (uint32_t img = (tag == 'I' || m.old_kind == ETLG_OLD_NONE) ? 1u : 0u; img < (tag == 'D' ? 1u : 2u); img++) {
        const bool is_new = img == 1;
        int mode;
        if (is_new) { fixed += s.row_full; mode = tag == 'U' ? 3 : 0; }
        else if (m.old_kind == ETLG_OLD_FULL) { fixed += s.row_full; mode = 0; }
        else { fixed += s.row_key; mode = m.old_n == s.n_ident ? 1 : m.old_n == s.n_cols ? 2 : -1; }
        if (mode >= 0) heap += tuple_heap_bytes(p, s, is_new ? m.new_t : m.old_t, is_new ? m.new_n : m.old_n, mode, over);
      }
      break;
    }
    case 'T': {
      if (check_txn && !tx.in_txn) { record_error(p, f, RK_TXN, ETLG_E_TXN_STATE); break; }
      const uint32_t nrel = ld_be32(b);
      uint32_t owned = 0;
      for (uint32_t i = 0; i < nrel; i++) {
        const uint32_t rel = ld_be32(b + 5 + 4 * i);
        const int ti = find_table(p, rel);
        if (!should_apply(p, ti, rel, tx.final_lsn)) continue;
        const int slot = cache_slot_before(p, ti, f);
        if (slot < 0) { record_error(p, f, RK_SCHEMA, (uint32_t)(-slot)); owned = 0; break; }
        owned++;
      }
      if (owned) { emit = 1; fixed = 8 * owned; }
      break;
    }
    default: break;  // 'k', 'O', 'Y'
  }
}

// The transaction-state checks of size_frame for a frame that was sized early.
DEV void txn_check_frame(const DecParams& p, const FrameView& v, const TxnCtx& tx) {
  const uint32_t tag = v.tag;
  if (tag == 'C') {
    if (!tx.in_txn) record_error(p, v.f, RK_TXN, ETLG_E_TXN_STATE);
    else if (ld_be64(v.fr + kBodyOff + 1) != tx.final_lsn) record_error(p, v.f, RK_TXN, ETLG_E_COMMIT_LSN);
  } else if (tag == 'R' || tag == 'I' || tag == 'U' || tag == 'D' || tag == 'T') {
    if (!tx.in_txn) record_error(p, v.f, RK_TXN, ETLG_E_TXN_STATE);
  }
}

// Decodes one emitting frame into the arena at (ev_idx, fx_off, hp_off).
// `pu` (optional): the same parameters with GLOBAL side-table pointers, enabling the
// wave-uniform INSERT path; `over`: the frame bytes may be over-read by up to 7 bytes.
DEV void write_frame(const DecParams& p, const FrameView& v, const TxnCtx& tx, const RowMsg& m, int row_slot,
                     uint64_t ev_idx, uint64_t fx_off, uint64_t hp_off, const DecParams* pu = nullptr, bool over = false) {
  const uint32_t f = v.f, tag = v.tag;
  const u8* fr = v.fr;
  const u8* b = fr + kBodyOff;
  u8* body = p.fixed + fx_off;
  uint32_t flags = 0, table = 0, slot_id = 0;
  uint64_t commit_lsn = tx.final_lsn;
  switch (tag) {
    case 'B':  // parse_event_from_begin_message, codec/event.rs:303-316
      commit_lsn = ld_be64(b);
      st64((uint32_t*)body, ld_be64(b + 8));
      table = ld_be32(b + 16);
      break;
    case 'C':  // parse_event_from_commit_message, codec/event.rs:322-336
      flags = b[0];
      commit_lsn = ld_be64(b + 1);
      st64((uint32_t*)body, ld_be64(b + 9));
      st64((uint32_t*)body + 2, ld_be64(b + 17));
      break;
    case 'R': {
      table = ld_be32(b);
      const DevEpoch* ep = epoch_at(p, find_table(p, table), f);
      slot_id = ep ? slot_host_id(p, ep->slot) : 0;
      break;
    }
    case 'T': {  // parse_event_from_truncate_message, codec/event.rs:533-547
      const uint32_t nrel = ld_be32(b);
      flags = b[4];
      uint32_t k = 0;
      for (uint32_t i = 0; i < nrel; i++) {
        const uint32_t rel = ld_be32(b + 5 + 4 * i);
        const int ti = find_table(p, rel);
        if (!should_apply(p, ti, rel, tx.final_lsn)) continue;
        const int sl = cache_slot_before(p, ti, f);
        ((uint32_t*)body)[2 * k] = rel; ((uint32_t*)body)[2 * k + 1] = slot_host_id(p, sl);
        k++;
      }
      table = k;
      break;
    }
    default: {  // I / U / D
      table = m.rel_id;
      const int sl = row_slot >= 0 ? row_slot : cache_slot_before(p, find_table(p, m.rel_id), f);
      slot_id = slot_host_id(p, sl);
      uint32_t hcur = (uint32_t)hp_off;
      uint32_t err = 0;
      bool partial = false;
      const uint32_t sl_u = __builtin_amdgcn_readfirstlane((uint32_t)sl);
      const uint32_t n_u = __builtin_amdgcn_readfirstlane(m.new_n);
      if (p.flags & 0x400u) { /* profiling ablation: no row decode at all */ }
      else if (p.flags & 0x800u) { if (pu && __all(tag == 'I' && (uint32_t)sl == sl_u && m.new_n == n_u)) err = write_full_row_uniform(*pu, sl_u, m.new_t, n_u, body, hcur, over); /* ablation: uniform waves only */ }
      else if (pu && __all(tag == 'I' && (uint32_t)sl == sl_u && m.new_n == n_u)) {
        err = write_full_row_uniform(*pu, sl_u, m.new_t, n_u, body, hcur, over);
      } else {
        const DevSlot& s = p.slots[sl];
        const uint32_t old_sz = m.old_kind == ETLG_OLD_FULL ? s.row_full : m.old_kind == ETLG_OLD_KEY ? s.row_key : 0;
        if (tag != 'I') flags = m.old_kind;
        // image 0 = old / key tuple (U, D), image 1 = new tuple (I, U): one call site for both
        for (uint32_t img = (tag == 'I' || m.old_kind == ETLG_OLD_NONE) ? 1u : 0u; img < (tag == 'D' ? 1u : 2u) && !err; img++) {
          const bool is_new = img == 1;
          const uint32_t mode = is_new ? (tag == 'U' ? (uint32_t)ROW_UPDATE : (uint32_t)ROW_FULL)
                                       : (m.old_kind == ETLG_OLD_KEY ? (uint32_t)ROW_KEY : (uint32_t)ROW_FULL);
          err = write_row(p, s, mode, is_new ? m.new_t : m.old_t, is_new ? m.new_n : m.old_n,
                          is_new ? body + old_sz : body, m.old_kind, body, hcur, partial, over);
        }
      }
      if (partial) flags |= ETLG_FLAG_PARTIAL;
      if (err) { record_error(p, f, RK_DECODE, err); return; }
      break;
    }
  }
  if (p.flags & 0x200u) return;  // profiling ablation (no header stores)
  p.ev_kind[ev_idx] = (u8)tag;
  p.ev_flags[ev_idx] = (u8)flags;
  p.ev_table[ev_idx] = table;
  p.ev_slot[ev_idx] = slot_id;
  p.ev_start[ev_idx] = ld_be64(fr + 6);  // wal_start (apply.rs:2039)
  p.ev_commit[ev_idx] = commit_lsn;
  p.ev_ord[ev_idx] = tx.ord;
  p.ev_body[ev_idx] = fx_off;
}

// classify from a frame pointer (LDS or global) + its sidecar extent
DEV uint32_t classify_ptr(const u8* fr, uint32_t flen) {
  if (flen < kHdr + 1 || fr[0] != 'd' || ld_be32(fr + 1) + 1u != flen) return FT_BAD;
  const uint32_t outer = fr[5];
  if (outer == 'k') return flen >= kHdr + 18 ? 'k' : FT_BAD;
  if (outer != 'w' || flen < kBodyOff) return FT_BAD;
  const uint32_t tag = fr[kTagOff];
  switch (tag) {
    case 'B': return flen >= kBodyOff + 20 ? tag : FT_BAD;  // u64 final_lsn, i64 ts, u32 xid
    case 'C': return flen >= kBodyOff + 25 ? tag : FT_BAD;  // i8 flags, u64, u64, i64
    case 'O': case 'Y': case 'R': case 'M': case 'I': case 'U': case 'D': case 'T': return tag;
    default: return FT_BAD;
  }
}

DEV uint32_t classify_frame(const DecParams& p, uint32_t f) {
  const uint32_t o0 = p.offs[f], o1 = p.offs[f + 1];
  if (o1 <= o0 || o1 > p.in_len) return FT_BAD;
  return classify_ptr(p.in + o0, o1 - o0);
}

}  // namespace etlg
