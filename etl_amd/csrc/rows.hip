// Row-synchronous single-pass decode kernel for gfx950 — the variable-length path since round 6.
//
// k_cells (cells.hip) slices a tile's tuples into a cell table and then visits the table column by column: ~25 k
// instructions per 64-frame tile, a third of them the per-visit skeleton (frame state, slot / column records, arena
// addresses looked up per LANE), a sizing pass that parses numerics a second time, a per-frame prefix over all virtual
// columns, and text moved four bytes per lane. Here a tile is:
//
//   P0  all waves: stage the tile's bytes + side tables into LDS
//   P1  spine wave, lane = frame: envelope, tag, transaction scan, ownership + schema slot, FIXED-arena bytes per frame
//       (they depend on the message heads only) -> wave scan; the fixed-arena and transaction aggregates are published at
//       once and resolved by two other waves during W, so that D knows every row's final place in the arena
//   W   spine wave: a lean structural walk (tag, length, bounds; ~35 instructions per step) that leaves the position of
//       every cell in a 16-bit table [virtual column][frame]; then the frames are sorted into GROUPS of one (schema slot,
//       image kind, cell count): inside a group, cell k of every frame belongs to the same column
//   D   all waves pull (group, column) tasks: the column record is wave-uniform (scalar loads, a scalar class switch — the
//       per-schema program the host would otherwise have to compile), lanes = the group's frames, heaviest classes first.
//       Fixed-width values go straight to their arena slots; cells that reach the heap (text, numeric, bytea, deferred
//       classes) are sized and noted in a small table [heap rank][frame]; 2-bit cell states collect in LDS
//   S   spine: per-frame prefix over its heap cells (heap rows only), wave scan of (events, heap dwords), look-back
//   H   spine: numerics / bytea emitted at their final heap positions, heap references patched into the slots, toast
//       cells aliased, row state words and event headers stored —
//   C   — while the other waves copy the text of String / deferred cells by 16-byte chunks dealt out densely over the
//       lanes (a chunk's owner cell by binary search over the row's chunk prefix sums), validated as UTF-8 on the way
//
// What the kernel does not cover it hands back (DevResult.fused_fail bit 4, "rows gave up": the host decodes the batch
// again with k_cells / k_fused): a tile whose bytes do not fit its LDS window, a
// frame with more than 32 KiB of heap entries or longer than 64 KiB, more than 16 groups in one tile. Errors are recorded
// at their frame like everywhere else and send the batch to the multi-pass kernels for the exact cut (the code recorded
// here only has to be SOME error of that frame).
#define ETLG_FLOAT_CALL static __device__ __attribute__((noinline))
#define ETLG_DBG_WORD dbg_u
#define ETLG_TSTAMP_WHO (spine && lane == 0)
#define ETLG_TEMPORAL_SWAR 1   // timestamps / whole-hour offsets decided from registers (codec.hip.h)
#include "lookback.hip.h"
#include "utf8_swar.h"

namespace etlg {

constexpr int RNW = 4;          // waves per tile
constexpr uint32_t kRowsGaveUp = 16u;
constexpr uint32_t kRowsMaxGroups = 16;
// heap-cell table entry: window position of the text (17 bits) | heap dwords of the cell, later its heap offset inside the frame (13 bits) | kind (2 bits)
enum : uint32_t { HK_NONE = 0, HK_COPY = 1, HK_NUMERIC = 2, HK_BYTEA = 3 };
constexpr uint32_t kRowsHeapMaxDw = 0x1FFFu;

DEV bool rows_heap_class(uint32_t cls) {
  return !(cls == ETLG_TC_BOOL || cls == ETLG_TC_I16 || cls == ETLG_TC_I32 || cls == ETLG_TC_I64 || cls == ETLG_TC_U32 || cls == ETLG_TC_UUID ||
           cls == ETLG_TC_DATE || cls == ETLG_TC_TIME || cls == ETLG_TC_TIMETZ || cls == ETLG_TC_TIMESTAMP || cls == ETLG_TC_TIMESTAMPTZ);
}

#ifndef ETLG_ROWS_MINBLOCKS
#define ETLG_ROWS_MINBLOCKS 4
#endif

template <int NW>
__global__ __launch_bounds__(NW * 64, ETLG_ROWS_MINBLOCKS) void k_rows(DecParams pg, FusedParams q) {
  ETLG_DYNAMIC_LDS(smem);
  __shared__ uint32_t s_offs[64 + 1];
  __shared__ uint32_t fr_hp[64];      // heap offset of the frame's first entry (absolute, bytes)
  __shared__ uint32_t fr_base[64];    // window offset of the frame's first byte
  __shared__ uint32_t fr_row[64];     // the frame's body inside the tile's piece of the fixed arena, bytes
  __shared__ uint32_t fr_osz[64];     // bytes of its old / key row
  __shared__ uint32_t row_next[128];       // ... and the next unclaimed step (chunk index) of every row
  __shared__ uint32_t row_chunks[128];     // 16-byte chunks of text to copy per row of the heap-cell table (D adds them up, C deals the rows out by them)
  __shared__ uint8_t own_mark[NW][64];     // text copy: which cell starts at a chunk position of the current step
  __shared__ uint32_t fr_flags[64];   // bit 0: a cell failed to decode; bit 1: the frame is an Update; bit 2: beyond what the kernel covers
  __shared__ uint64_t g_mem[kRowsMaxGroups];
  __shared__ uint32_t s32[16];
  __shared__ uint64_t s64[12];
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  if (q.clear_words) {  // descriptors are double buffered: this launch clears the buffer the next batch will use
    const uint32_t per = (q.clear_words + gridDim.x - 1) / gridDim.x;
    for (uint32_t i = tid; i < per; i += NW * 64) { const uint32_t w = blockIdx.x * per + i; if (w < q.clear_words) q.d_clear[w] = 0; }
  }
  if (!(pg.flags & 16u) && !load_carry(pg)) return;  // ASYNC chain: the state the batch before this one left (flags bit 4: read late, by the tiles that need it)
  DecParams p = pg;
  uint32_t dbg_u, cf, maxc;
  ETLG_SCALAR_COPY(dbg_u, q.dbg); ETLG_SCALAR_COPY(cf, q.blk); ETLG_SCALAR_COPY(maxc, q.maxc);
  const uint32_t tile = blockIdx.x;
  const uint32_t role = ((tid >> 6) + tile) & (uint32_t)(NW - 1);   // the roles rotate with the tile (see cells.hip: wave w of a workgroup sits on SIMD w)
  const bool spine = role == 0;
  if ((dbg_u & 8) && spine && lane == 0) s64[7] = clock64();
  uint32_t* fail = &pg.res->fused_fail;
  // ---- P0: side tables, offsets, staging
  SideRegs side;
  side_load<NW * 64>(p, true, (uint32_t*)smem, tid, side);
  // rows of the heap-cell table: the heap-class columns a key / old image can hold (identity columns), then those of a full row
  const uint32_t maxh = q.rows_maxh, maxh_old = q.rows_maxh_old, R = maxh_old + maxh;
  // dynamic LDS: side tables | heap-cell table (u32 [R][cf]) | its slots (u16 [R][cf]: dword offsets inside the tile's piece of the fixed arena)
  //              | 2-bit cell states (u32 [2 SW][cf]: one word per 16 columns and image) | toast masks | cell positions (u16 [2 maxc][cf]) | task list | staging window
  const uint32_t SW = (maxc + 15u) >> 4;
  uint32_t* const htab = (uint32_t*)(smem + q.side_bytes);
  uint16_t* const hslot = (uint16_t*)(htab + R * cf);
  const uint32_t htab_bytes = (R * cf * 6u + 15u) & ~15u;
  uint32_t* const sttab = (uint32_t*)((u8*)htab + htab_bytes);
  const uint32_t st_bytes = (2u * SW * cf * 4u + 15u) & ~15u;
  const uint32_t TW = (maxc + 31u) >> 5;
  uint32_t* const toast = (uint32_t*)((u8*)sttab + st_bytes);   // new-image columns sent as 'u': u32 [TW][cf]
  const uint32_t toast_bytes = (TW * cf * 4u + 15u) & ~15u;
  uint16_t* const ctab = (uint16_t*)((u8*)toast + toast_bytes);
  const uint32_t ctab_bytes = (2u * maxc * cf * 2u + 15u) & ~15u;
  // the task list: {cell index | group << 8 | image shape << 12 | new image << 14, the column's DevCol record}, heavy classes first inside a group
  uint4* const tasks = (uint4*)((u8*)ctab + ctab_bytes);
  const uint32_t max_tasks = 4u * maxc + 32u;   // (a tile usually holds one or two groups; one that needs more tasks is handed back)
  u8* const stage = (u8*)tasks + max_tasks * 16u;
  const uint32_t f0 = tile * cf;
  const uint32_t nt = pg.nframes - f0 < cf ? pg.nframes - f0 : cf;
  const ETLG_CONST_AS uint32_t* offs_c = (const ETLG_CONST_AS uint32_t*)(uintptr_t)pg.offs;
  const uint32_t span0 = offs_c[f0], span1 = offs_c[f0 + nt];
  const uint32_t my_o = tid <= nt ? pg.offs[f0 + tid] : 0u;
  const uint32_t a0 = span0 & ~15u;
  const uint32_t used = q.side_bytes + htab_bytes + st_bytes + toast_bytes + ctab_bytes + max_tasks * 16u;
  const uint32_t wcap = q.lds_bytes > used ? q.lds_bytes - used : 0u;
  const bool window_ok = q.in_aligned && span1 > span0 && span1 <= pg.in_len && span1 - a0 + 16 <= (1u << 17) && (uint64_t)(span1 - a0) + 16 <= wcap;
  if (window_ok) {
    const uint32_t full_end = a0 + ((span1 - a0) & ~15u);
    stage_chunks<NW * 64>(pg.in, stage, a0, full_end, tid);
    for (uint32_t c = full_end + tid; c < span1; c += NW * 64) stage[c - a0] = pg.in[c];
  }
  side_store<NW * 64>((uint32_t*)smem, tid, side);
  if (tid <= nt) s_offs[tid] = my_o;
  {  // the heap-cell table starts empty, the cell states as VALUE, no cell as toast
    uint4* z = (uint4*)htab;
    const uint32_t n16 = (htab_bytes + st_bytes + toast_bytes) >> 4;
    for (uint32_t i = tid; i < n16; i += NW * 64) z[i] = make_uint4(0, 0, 0, 0);
  }
  if (tid < 64) fr_flags[tid] = 0;
  if (tid < 128) { row_chunks[tid] = 0; row_next[tid] = 0; }
  if (tid == 0) { s64[8] = 0; s64[9] = 0; s32[2] = 0; s32[3] = 0; s32[4] = 0; }
  __syncthreads();
  TSTAMP(0);
  bool lane_ok = true;
  if (tid < nt) {
    const uint32_t o0 = s_offs[tid], o1 = s_offs[tid + 1];
    lane_ok = o1 <= o0 || o1 > pg.in_len || (o0 >= span0 && o1 <= span1);
  }
  const bool tile_ok = __syncthreads_and(lane_ok ? 1 : 0) && window_ok;
  if (!tile_ok && tid == 0) {   // (the tile still takes part in the look-backs, with nothing to report)
    atomicOr(fail, kRowsGaveUp | 0x100u);
    atomicMax(&pg.res->dbg_t[11], (unsigned long long)(span1 - a0) + 32ull);   // what its window would have had to hold: the host sizes the next attempt by it
  }
  const u8* const base = stage;
  const uint32_t b0 = a0;

  // ================= P1 (spine): heads, transaction scan, slots, fixed sizes
  const bool live = spine && lane < nt && tile_ok;
  const uint32_t f = f0 + lane;
  FrameView v{f, 0, base, base};
  uint32_t rel_id = 0, old_kind = ETLG_OLD_NONE, n_old = 0, n_new = 0, vbytes = 0, o0 = 0;
  uint32_t c = 0, e = 0, fb = 0;  // walk cursor / frame end / frame start, as window offsets
  uint32_t ok = 0;                // 1: a row frame that is still well formed
  uint32_t derr = 0, gave = 0;
  bool wire_ok = true, isrow = false;
  uint32_t cnt = 0, mark = 0, seg_in = 0, pm = 0, tot_cnt = 0, tot_mark = 0;
  int slot = -1;
  uint32_t s_ncols = 0, s_nident = 0, s_cb = 0;
  uint32_t emit = 0, fixed = 0, heap = 0, old_sz = 0, x_fx = 0;
  uint64_t pay[3] = {0, 0, 0};
  TxnCtx tx{true, 0, 0};
  uint32_t bc = 0, bm = 0;
  uint64_t carried_lsn = 0, start_ord = 0;
  uint32_t tot_f = 0;
  if (spine) {
    ETLG_WAVE_PRIO(3);
    if (live) {
      o0 = s_offs[lane];
      const uint32_t o1 = s_offs[lane + 1];
      if (o1 > o0 && o1 <= pg.in_len) {
        v.fr = base + (o0 - b0);
        v.e = base + (o1 - b0);
        v.tag = classify_ptr(v.fr, o1 - o0);
      }
      const uint32_t tag = v.tag;
      isrow = tag == 'I' || tag == 'U' || tag == 'D';
      if (!isrow) {
        RowMsg dummy;
        wire_ok = frame_structure(v, dummy, true);
      } else {
        fb = (uint32_t)(v.fr - base);
        c = fb + kBodyOff; e = (uint32_t)(v.e - base);
        if (e - fb > 0xFFFFu) gave = 0x400;   // cell positions are 16-bit offsets from the frame's first byte
        wire_ok = e >= c + 5;
        if (wire_ok) {
          rel_id = ld_be32(base + c); c += 4;
          // the first image header: 'K' | 'O' | 'N', i16 column count
          const uint64_t head = ldu64(base + c);
          const uint32_t t = (uint32_t)head & 0xFFu;
          const uint32_t cnt16 = (((uint32_t)head >> 8) & 0xFFu) << 8 | (((uint32_t)head >> 16) & 0xFFu);
          const bool is_old = (t == 'K') | (t == 'O');
          const bool hdr_ok = (e - c >= 3) & !(cnt16 & 0x8000u) & (tag == 'I' ? t == 'N' : tag == 'D' ? is_old : (is_old | (t == 'N')));
          wire_ok = hdr_ok;
          if (hdr_ok) {
            c += 3; ok = gave ? 0u : 1u;
            if (is_old) { old_kind = t == 'K' ? (uint32_t)ETLG_OLD_KEY : (uint32_t)ETLG_OLD_FULL; n_old = cnt16; }
            else n_new = cnt16;
          }
        }
      }
      if (consumes_ordinal(tag)) cnt = 1;
      if (tag == 'B') { cnt |= 0x80000000u; mark = ((o0 + 1) << 1) | 1; }
      if (tag == 'C') mark = (o0 + 1) << 1;
    }
    {  // wave-level transaction scan (a tile's frames live in one wave)
      const uint32_t ic = wave_scan_incl(cnt, [](uint32_t a, uint32_t b) { return seg_combine(a, b); }, 0u);
      const uint32_t im = wave_scan_max(mark);
      pm = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)im, 0x138, 0xF, 0xF, false);  // previous lane's value, 0 into lane 0
      seg_in = ic;
      tot_cnt = wave_last(ic); tot_mark = wave_last(im);
    }
    // ownership + schema slot, fixed-arena bytes of the frame
    if (isrow && wire_ok) {
      const int ti = find_table(p, rel_id);
      if (should_apply(p, ti, rel_id, 0)) {   // (no table is in SyncDone state on this path: the host sends those batches to k_cells)
        slot = cache_slot_before(p, ti, f);
        if (slot < 0) { record_error(pg, f, RK_SCHEMA, (uint32_t)(-slot)); slot = -1; derr = 1; }
        else if (p.slots[slot].n_cols > maxc) { gave = 0x800; slot = -1; }
      }
      if (slot >= 0) {
        const DevSlot& s = p.slots[slot];
        s_ncols = s.n_cols; s_nident = s.n_ident; s_cb = s.cols_base;
        emit = 1;
        if (old_kind != ETLG_OLD_NONE) { old_sz = old_kind == ETLG_OLD_KEY ? s.row_key : s.row_full; fixed = old_sz; }
        if (v.tag != 'D') fixed += s.row_full;
      }
    } else if (live && !isrow) {
      RowMsg dummy{};
      int rs = -1;
      size_frame(p, v, tx, wire_ok, dummy, emit, fixed, heap, pay, rs, false, true);   // (records the wire error of a malformed frame)
    }
    const uint32_t ifx = wave_scan_add(fixed >> 2);
    tot_f = wave_last(ifx);
    x_fx = ifx - (fixed >> 2);
    if (tot_f > 0xFFFFu) {   // heap cells name their slots by 16-bit dword offsets inside the tile's piece of the arena
      gave = 0x200; tot_f = 0; emit = 0; fixed = 0; ok = 0; x_fx = 0; slot = -1;
    }
    if (lane < cf) { fr_base[lane] = fb; fr_row[lane] = x_fx << 2; fr_osz[lane] = old_sz; if (isrow && v.tag == 'U') fr_flags[lane] = 2u; }
    if (lane == 0) { s64[1] = tot_f; s64[2] = ((uint64_t)seg_pack30(tot_cnt) << 32) | tot_mark; }
    ETLG_WAVE_PRIO(0);
  }
  __syncthreads();
  TSTAMP(1);
  // the two look-backs whose aggregates are known from the heads alone run beside the walk
  if (role == 1 % NW && NW > 1) { const uint64_t b = lookback<OpAdd>(q.d_outb, q.d_outb + q.ntiles, tile, s64[1], 0, fail); if (lane == 0) s64[5] = b; }
  // (the transaction fold only: what it needs of the state carried into the batch is fetched at the very end of the tile — H — because
  // with ETLG_F_ASYNC on two streams that state may have to be waited for, and a tile that waits before it has published its event /
  // heap aggregate would stall every tile behind it while the batch before still needs their CU slots)
  if (role == 2 % NW && NW > 2) { const uint64_t ex = lookback<OpTxn>(q.d_txn, q.d_txn + q.ntiles, tile, s64[2], 0ull, fail); if (lane == 0) s64[10] = ex; }

  if (spine) {
    if (NW == 1) { const uint64_t b = lookback<OpAdd>(q.d_outb, q.d_outb + q.ntiles, tile, s64[1], 0, fail); if (lane == 0) s64[5] = b; }
    if (NW <= 2) { const uint64_t ex = lookback<OpTxn>(q.d_txn, q.d_txn + q.ntiles, tile, s64[2], 0ull, fail); if (lane == 0) s64[10] = ex; }
    // ================= W (spine): the structural walk
    ETLG_WAVE_PRIO(3);
    // `lim` cells of image `vimg` (0 old / key, 1 new) for every lane that is still going: 'n' | 'u' | ('t' | 'b') i32 len bytes.
    // One step per cell index, no data-dependent branches; a lane that meets a malformed cell stops (ok = 0).
    auto walk_cells = [&](uint32_t lim, uint32_t vimg) {
      // (integer flags and one table lookup for the four cell tags: as predicates every test lived in a lane-mask register pair and the
      // loop spent a third of its instructions combining those)
      constexpr uint32_t kValid = (1u << ('b' - 'b')) | (1u << ('n' - 'b')) | (1u << ('t' - 'b')) | (1u << ('u' - 'b'));
      constexpr uint32_t kValue = (1u << ('b' - 'b')) | (1u << ('t' - 'b'));
      uint16_t* const row0 = ctab + vimg * maxc * cf + lane;
      const uint32_t lim_k = lim < maxc ? lim : maxc;   // cells beyond the widest slot are walked, not noted (their frame decodes against no slot)
      for (uint32_t k = 0;; k++) {
        const uint32_t act = (k < lim) ? ok : 0u;
        if (!__ballot(act != 0)) break;
        const uint64_t head = ldu64(base + c);  // the window has 16 spare bytes past any frame
        const uint32_t ti = ((uint32_t)head & 0xFFu) - 'b';
        const uint32_t sh5 = ti < 32u ? ti : 31u;
        const uint32_t valid = (kValid >> sh5) & (ti < 32u ? 1u : 0u), is_val = (kValue >> sh5) & valid;
        const uint32_t len = __builtin_bswap32((uint32_t)(head >> 8)) & (0u - is_val);
        const uint32_t room = e - c;
        const uint32_t need = 1u + (is_val << 2) + len;                       // the bytes the cell takes
        const uint32_t go = (valid & act & (len <= room ? 1u : 0u)) & (need <= room ? 1u : 0u);   // (len first: need may wrap for a length near 2^32)
        if (go && k < lim_k) row0[k * cf] = (uint16_t)(c - fb);
        ok = act & (go ^ 1u) ? 0u : ok;
        vbytes += len & (0u - go);
        c += need & (0u - go);
      }
    };
    walk_cells(old_kind != ETLG_OLD_NONE ? n_old : 0u, 0u);
    {  // an Update goes on to its new image's header
      const bool need = (ok != 0) & (v.tag == 'U') & (old_kind != ETLG_OLD_NONE);
      const uint64_t head = ldu64(base + c);
      const uint32_t t = (uint32_t)head & 0xFFu;
      const uint32_t cnt16 = (((uint32_t)head >> 8) & 0xFFu) << 8 | (((uint32_t)head >> 16) & 0xFFu);
      const bool hdr_ok = (e - c >= 3) & (t == 'N') & !(cnt16 & 0x8000u);
      if (need) { if (hdr_ok) { c += 3; n_new = cnt16; } else ok = 0; }
    }
    walk_cells(v.tag != 'D' ? n_new : 0u, 1u);
    if (isrow) wire_ok = ok != 0;
    if (live && isrow && !wire_ok) record_error(pg, f, RK_WIRE, ETLG_E_WIRE);
    TSTAMP(2);
    // ---- groups: frames of one (slot, image shape, cell count); convert_tuple_to_row / normalize_key_tuple_to_row's shape checks
    //      (codec/event.rs:559-565, 889-922) per frame: 0 full / update row, 1 dense key tuple, 2 full-width key tuple, 3 = a shape error
    const uint32_t mo = old_kind == ETLG_OLD_KEY ? (s_nident == 0 ? 3u : n_old == s_nident ? 1u : n_old == s_ncols ? 2u : 3u) : (n_old == s_ncols ? 0u : 3u);
    const uint32_t mn = n_new == s_ncols ? 0u : 3u;
    // ... and, group by group, the task list: inside every group the heavy classes first (temporal / uuid / numeric / float), so that the
    // waves dealing the list out among themselves finish close to each other. An entry carries the column's record.
    uint32_t ng = 0, nq = 0;
    for (uint32_t img1 = 0; img1 < 2; img1++) {
      const uint32_t myn = img1 ? n_new : n_old, mym = img1 ? mn : mo;
      uint32_t pend = (ok != 0 && slot >= 0 && !derr && (img1 ? v.tag != 'D' : old_kind != ETLG_OLD_NONE)) ? 1u : 0u;
      for (;;) {
        const unsigned long long m = __ballot(pend != 0);
        if (!m) break;
        const int leader = __builtin_ctzll(m);
        const uint32_t slot_u = (uint32_t)__builtin_amdgcn_readlane(slot, leader);
        const uint32_t n_u = (uint32_t)__builtin_amdgcn_readlane((int)myn, leader);
        const uint32_t mode_u = (uint32_t)__builtin_amdgcn_readlane((int)mym, leader);
        const uint32_t cb_u = (uint32_t)__builtin_amdgcn_readlane((int)s_cb, leader);
        const bool in = pend && (uint32_t)slot == slot_u && myn == n_u && mym == mode_u;
        const unsigned long long mem = __ballot(in);
        if (in) pend = 0;
        if (mode_u == 3) { if (in) derr = 1; continue; }   // (the frame fails whatever its cells hold)
        if (n_u == 0) continue;
        if (ng >= kRowsMaxGroups || nq + n_u > max_tasks) { if (in) gave = 0x1000; continue; }
        if (lane == 0) g_mem[ng] = mem;
        const uint32_t tag16 = (ng << 8) | (mode_u << 12) | (img1 << 14);
        for (uint32_t k0 = 0; k0 < n_u; k0 += 64) {
          const uint32_t k = k0 + lane;
          uint32_t heavy = 0, light = 0;
          uint4 ent = make_uint4(0, 0, 0, 0);
          if (k < n_u) {
            const uint32_t* cw = (const uint32_t*)(p.cols + cb_u);   // (the LDS copy of the column records)
            static_assert(sizeof(DevCol) == 12, "descriptor words below");
            const uint32_t ci = mode_u == 1 ? cw[3 * k] >> 24 : k;   // DevCol.key_col of record k: cell k of a dense key tuple
            ent = make_uint4(k | tag16, cw[3 * ci], cw[3 * ci + 1], cw[3 * ci + 2]);
            const uint32_t cls = ent.y & 0xFFu;
            const bool hv = (cls >= ETLG_TC_DATE && cls <= ETLG_TC_UUID) || cls == ETLG_TC_NUMERIC || cls == ETLG_TC_F32 || cls == ETLG_TC_F64;
            const bool take = !(mode_u == 2 && !((ent.y >> 16) & 0xFFu));   // (full-width key tuple: only the identity columns are read)
            heavy = (take && hv) ? 1u : 0u; light = (take && !hv) ? 1u : 0u;
          }
          const unsigned long long mh = __ballot(heavy != 0), ml = __ballot(light != 0);
          const unsigned long long lt = (1ull << lane) - 1ull;
          const uint32_t nh = (uint32_t)__builtin_popcountll(mh);
          if (heavy) tasks[nq + (uint32_t)__builtin_popcountll(mh & lt)] = ent;
          if (light) tasks[nq + nh + (uint32_t)__builtin_popcountll(ml & lt)] = ent;
          nq += nh + (uint32_t)__builtin_popcountll(ml);
        }
        ng++;
      }
    }
    TSTAMP(9);
    if (lane == 0) { s32[2] = nq; s32[3] = ng; }
    ETLG_WAVE_PRIO(0);
  }
  __syncthreads();
  TSTAMP(3);

  // ================= D (all waves): (group, column) tasks
  const uint64_t pre_fx = s64[5] << 2;   // (resolved beside the walk)
  const bool fixed_fits = pre_fx + (s64[1] << 2) <= pg.fixed_cap;
  {
    const uint32_t ntasks = fixed_fits ? s32[2] : 0u;
    // (the frame's constants once per wave; the tasks are dealt out round robin — they were ordered heavy first — so that a wave knows its
    // next task without asking: one LDS round trip per task for the record instead of an atomic, a list entry, three group words and two
    // dependent scalar loads from global memory)
    const uint32_t my_base = fr_base[lane], my_row = fr_row[lane], my_osz = fr_osz[lane], my_upd = fr_flags[lane] & 2u;
    for (uint32_t tk = tid >> 6; tk < ntasks; tk += NW) {
      const uint4 ent = tasks[tk];
      const uint32_t e0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent.x);
      const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent.y), w1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent.z),
                     w2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent.w);
      const uint32_t k = e0 & 0xFFu, g = (e0 >> 8) & 0xFu, kmode = (e0 >> 12) & 3u;
      const bool img1 = ((e0 >> 14) & 1u) != 0;
      const uint64_t mem = g_mem[g];
      const uint32_t cls = w0 & 0xFFu, nullable = (w0 >> 8) & 0xFFu;
      const uint32_t off = kmode ? (w1 >> 16) : (w1 & 0xFFFFu);
      const uint32_t kout = kmode ? (w2 & 0xFFFFu) : k;
      const uint32_t hr = kmode ? (w2 >> 24) : ((w2 >> 16) & 0xFFu);
      const bool on = ((mem >> lane) & 1ull) != 0;
      const uint32_t pos0 = on ? my_base + ctab[((img1 ? maxc : 0u) + k) * cf + lane] : 0u;
      const uint32_t soff = my_row + (img1 ? my_osz : 0u) + off;   // the slot inside the tile's piece of the fixed arena
      uint32_t* const slotp = (uint32_t*)(pg.fixed + pre_fx + soff);
      const uint64_t head = ldu64(base + pos0);
      const uint32_t t = (uint32_t)head & 0xFFu;
      const uint32_t len = __builtin_bswap32((uint32_t)(head >> 8));
      const uint32_t pos = pos0 + 5;
      uint32_t st = ETLG_CELL_VALUE, bad = 0;
      bool entry = false;
      uint32_t cchunks = 0;   // 16-byte chunks of text the copy phase will move for this cell
      if (on && t == 't') {
        const u8* d = base + pos;
        if (rows_heap_class(cls)) {
          uint32_t kind = HK_COPY, nbytes = len;
          if (cls == ETLG_TC_NUMERIC) {
            NumShape ns;
            const bool okn = numeric_plain(d, len, ns) || numeric_scan(d, len, ns, true);
            kind = HK_NUMERIC;
            nbytes = okn ? 8 + 2 * ns.ngroups : 0u;
            if (!okn) bad = 1;
          } else if (cls == ETLG_TC_BYTEA) {
            if (len < 2) bad = 1;
            nbytes = len >= 2 ? (len - 2) >> 1 : 0u; kind = HK_BYTEA;
          } else if (cls == ETLG_TC_F32 || cls == ETLG_TC_F64) {
            uint64_t bits = 0;
            const int r = parse_float_fast(d, len, cls == ETLG_TC_F32, bits, true);
            if (r == 2) bad = 1;
            if (r == 0) { st64(slotp, bits); kind = HK_NONE; nbytes = 0; }
            else st = ETLG_CELL_DEFERRED;
          } else if (cls != ETLG_TC_STRING) st = ETLG_CELL_DEFERRED;   // json / arrays / classes without a codec: the source text
          const uint32_t hdw = (nbytes + 3u) >> 2;
          if (kind != HK_NONE && !bad) {
            if (hdw > kRowsHeapMaxDw || hr >= (img1 ? maxh : maxh_old)) bad = 4;   // (a full old image of a table whose replica identity is not FULL: no rows were set aside)
            else {
              const uint32_t r = (img1 ? maxh_old : 0u) + hr;
              htab[r * cf + lane] = pos | (hdw << 17) | (kind << 30);
              hslot[r * cf + lane] = (uint16_t)(soff >> 2);
              slotp[1] = nbytes;
              entry = true;
              if (kind == HK_COPY) cchunks = (len + 15u) >> 4;
            }
          }
        } else {
          uint32_t tmp[4] = {0, 0, 0, 0};
          uint32_t hdummy = 0;
          const uint32_t err = decode_text_cell<false>(cls, d, len, tmp, nullptr, hdummy, st, true);
          if (err) bad = 1;
          else {
            const uint32_t nw = slot_bytes(cls) >> 2;
            slotp[0] = tmp[0];
            if (nw > 1) slotp[1] = tmp[1];
            if (nw > 2) slotp[2] = tmp[2];
            if (nw > 3) slotp[3] = tmp[3];
          }
        }
      } else if (on && t == 'n') {
        if (!nullable) bad = 1;   // Required column missing from tuple (codec/event.rs:945-961)
        else slot_zero(slotp, cls);
        st = ETLG_CELL_NULL;
      } else if (on && t == 'u') {
        if (img1 && my_upd) { atomicOr(&toast[(k >> 5) * cf + lane], 1u << (k & 31u)); }   // resolved once the old image's heap references are final
        else bad = 1;              // a full row / key image cannot miss a value
      } else if (on) {
        bad = 1;                   // binary format
      }
      if (on && st && !bad) atomicOr(&sttab[((img1 ? SW : 0u) + (kout >> 4)) * cf + lane], st << (2 * (kout & 15u)));
      if (bad) atomicOr(&fr_flags[lane], bad);
      if (__ballot(entry)) {
        // (wave-uniform: some lane of the group left an entry) the row is marked present and its text chunks are added up
        const uint32_t tot = wave_last(wave_scan_add(cchunks));
        if (lane == 0) { atomicOr((unsigned long long*)&s64[img1 ? 9 : 8], 1ull << hr); if (tot) atomicAdd(&row_chunks[(img1 ? maxh_old : 0u) + hr], tot); }
      }
    }
  }
  __syncthreads();
  TSTAMP(4);

  uint32_t x_ev = 0, x_hp = 0;
  if (spine) {
    // ================= S (spine): per-frame prefix over its heap cells, wave scans, look-back
    ETLG_WAVE_PRIO(3);
    const uint64_t m_old = s64[8], m_new = s64[9];
    {
      const uint32_t fl = lane < cf ? fr_flags[lane] : 0u;
      if (fl & 1u) derr = 1;
      if (fl & 4u) gave = 0x2000;
    }
    uint32_t hdw = 0;
    for (uint32_t half = 0; half < 2; half++) {
      uint64_t m = half ? m_new : m_old;
      while (m) {
        const uint32_t hr = (uint32_t)__builtin_ctzll(m);
        m &= m - 1;
        const uint32_t idx = ((half ? maxh_old : 0u) + hr) * cf + (lane < cf ? lane : 0u);
        const uint32_t w = htab[idx];
        if ((w >> 30) != HK_NONE && lane < cf) {
          htab[idx] = (w & ~(kRowsHeapMaxDw << 17)) | ((hdw & kRowsHeapMaxDw) << 17);
          if (hdw > kRowsHeapMaxDw) gave = 0x4000;
          hdw += (w >> 17) & kRowsHeapMaxDw;
        }
      }
    }
    if (live && isrow) {
      if (wire_ok && derr) record_error(pg, f, RK_DECODE, ETLG_E_WIRE);
      pay[v.tag == 'I' ? 0 : v.tag == 'U' ? 1 : 2] = vbytes;
      if (slot >= 0) heap = hdw << 2;
    }
    { const uint32_t why = wave_last(wave_scan_incl(gave, [](uint32_t a, uint32_t b) { return a | b; }, 0u)); if (why && lane == 0) atomicOr(fail, kRowsGaveUp | why); }   // bits 8..: why (debugging aid)
    const uint32_t ie = wave_scan_add(emit), ih = wave_scan_add(heap >> 2);
    const uint32_t tot_e = wave_last(ie), tot_h = wave_last(ih);
    x_ev = ie - emit; x_hp = ih - (heap >> 2);
    const uint64_t agg = ((uint64_t)tot_e << 32) | tot_h;
    lookback_publish(q.d_outa, tile, agg);   // (at once: the tiles behind this one wait for it; the payload counters come after)
    const uint32_t a0p = wave_last(wave_scan_add((uint32_t)pay[0])), a1p = wave_last(wave_scan_add((uint32_t)pay[1])),
                   a2p = wave_last(wave_scan_add((uint32_t)pay[2]));
    if (lane == 0) {
      if (a0p) atomicAdd(&pg.res->pay_shard[tile & 31][0], (unsigned long long)a0p);
      if (a1p) atomicAdd(&pg.res->pay_shard[tile & 31][1], (unsigned long long)a1p);
      if (a2p) atomicAdd(&pg.res->pay_shard[tile & 31][2], (unsigned long long)a2p);
    }
    const uint64_t a = lookback_resolve<OpAdd2>(q.d_outa, q.d_outa + q.ntiles, tile, agg, 0, fail);
    const uint64_t pre_hp = (uint64_t)(uint32_t)a << 2;
    const bool hf = pre_hp + ((uint64_t)tot_h << 2) <= pg.heap_cap && pre_hp + ((uint64_t)tot_h << 2) <= 0xFFFFFFFFull;
    if (lane < cf) fr_hp[lane] = (uint32_t)(pre_hp + ((uint64_t)x_hp << 2));
    if (lane == 0) { s64[0] = agg; s64[4] = a; s32[0] = hf ? 1u : 0u; }
    ETLG_WAVE_PRIO(0);
  }
  __syncthreads();
  TSTAMP(5);
  const uint64_t pre_ev = s64[4] >> 32;
  const bool heap_fits = s32[0] != 0;
  const uint64_t m_old = s64[8], m_new = s64[9];

  if (spine) {
    // ================= H (spine): transaction context, numerics / bytea, heap references, toast, event headers
    ETLG_WAVE_PRIO(3);
    {
      // the second half of txn_lookback (lookback.hip.h): the state carried into the batch is not part of the fold; only a tile whose
      // prefix needs it reads it, and with DecParams.flags bit 4 (the batch before may still be running on the other decode stream)
      // waits for that batch's last tile first
      const uint64_t ex = s64[10];
      TxnStart ts;
      ts.seg = seg_unpack30((uint32_t)(ex >> 32));
      ts.mark = (uint32_t)ex;
      uint32_t c_in_txn = pg.in_txn;
      uint64_t c_final_lsn = pg.final_lsn;
      ts.ord = pg.next_ord;
      if ((pg.flags & 16u) && pg.carry && (ts.mark == 0u || !(ts.seg & 0x80000000u))) {   // wave-uniform
        for (uint32_t polls = 0;; polls++) {
          if (__hip_atomic_load(&pg.carry->carry_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) break;
          if (polls > (1u << 15)) { if (lane == 0) atomicOr(fail, 1u); break; }
          __builtin_amdgcn_s_sleep(8);
        }
        c_in_txn = __hip_atomic_load(&pg.carry->out_in_txn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c_final_lsn = __hip_atomic_load(&pg.carry->out_final_lsn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ts.ord = __hip_atomic_load(&pg.carry->out_next_ord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (ts.mark == 0u && c_in_txn) ts.mark = 1u;   // virtual Begin before frame 0
      ts.lsn = (ts.mark & 1u) ? (ts.mark == 1u ? c_final_lsn : ld_be64(pg.in + ((ts.mark >> 1) - 1) + kBodyOff)) : 0ull;
      bc = ts.seg; bm = ts.mark; carried_lsn = ts.lsn; start_ord = ts.ord;
      const uint32_t seg = seg_combine(bc, seg_in);
      const uint32_t last = bm > pm ? bm : pm;
      tx.in_txn = (last & 1u) != 0;
      tx.final_lsn = !tx.in_txn ? 0 : last == bm ? carried_lsn : ld_be64(base + (((last >> 1) - 1) - b0) + kBodyOff);
      const uint64_t cc = seg & 0x7FFFFFFFu;
      tx.ord = (seg & 0x80000000u) ? cc - 1 : start_ord + cc - 1;
      if (live && wire_ok) txn_check_frame(pg, v, tx);
    }
    const uint64_t ev_idx = pre_ev + x_ev;
    const uint64_t fx_off = pre_fx + ((uint64_t)x_fx << 2);
    if (lane == 0 && tile == q.ntiles - 1) {
      DevResult* r = pg.res;
      r->n_events = pre_ev + (s64[0] >> 32); r->fixed_bytes = pre_fx + (s64[1] << 2); r->heap_bytes = ((uint64_t)(uint32_t)s64[4] << 2) + ((uint64_t)(uint32_t)s64[0] << 2);
      r->n_frames = pg.nframes;
      const uint32_t sg = seg_combine(bc, tot_cnt);
      const uint32_t lm = bm > tot_mark ? bm : tot_mark;
      const bool it = (lm & 1u) != 0;
      r->out_in_txn = it;
      r->out_final_lsn = it ? (lm == bm ? carried_lsn : final_lsn_of_mark(pg, lm)) : 0;
      const uint64_t cc = sg & 0x7FFFFFFFu;
      r->out_next_ord = (sg & 0x80000000u) ? cc : start_ord + cc;
      carry_publish(r);
    }
    if (!(heap_fits && fixed_fits)) { if (emit) record_error(pg, f, RK_DECODE, ETLG_E_WIRE); emit = 0; }
    u8* const body = pg.fixed + fx_off;
    // 'u' cells of the new row: alias the aligned old value, else MISSING (codec/event.rs:962-974). The old row's slots were stored by
    // other waves of this workgroup before the barriers above and are read back past the L1 (agent scope).
    uint32_t flags = (isrow && v.tag != 'I') ? old_kind : 0u;
    for (uint32_t tw = 0; tw < TW; tw++) {
      uint32_t tbits = (emit && isrow && lane < cf) ? toast[tw * cf + lane] : 0u;
      if (!tbits) continue;
      const DevSlot& s = p.slots[slot];
      const DevCol* cols = p.cols + s.cols_base;
      while (tbits) {
        const uint32_t k = tw * 32u + (uint32_t)__builtin_ctz(tbits);
        tbits &= tbits - 1;
        const DevCol col = cols[k];
        uint32_t* dst = (uint32_t*)(body + old_sz + col.off_full);
        const bool from_full = old_kind == ETLG_OLD_FULL, from_key = old_kind == ETLG_OLD_KEY && col.identity;
        uint32_t cst;
        const uint32_t nw = slot_bytes(col.cls) >> 2;
        if (from_full || from_key) {
          uint32_t* src = (uint32_t*)(body + (from_full ? col.off_full : col.off_key));
          for (uint32_t w = 0; w < nw; w++) dst[w] = __hip_atomic_load(&src[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint32_t oi = from_full ? k : (uint32_t)col.key_index;
          cst = (sttab[(oi >> 4) * cf + lane] >> (2 * (oi & 15u))) & 3u;
          // an aliased cell that lives on the heap: its reference is being patched into the OLD slot by one of the copying waves right
          // now — the same value is computed here from the table instead of waiting for that store
          const uint32_t ohr = from_full ? (col.hr & 0xFFu) : (uint32_t)(col.hr >> 8);
          if (rows_heap_class(col.cls) && ohr != 0xFFu) {
            const uint32_t hw = htab[ohr * cf + lane];
            if ((hw >> 30) != HK_NONE) dst[0] = fr_hp[lane] + (((hw >> 17) & kRowsHeapMaxDw) << 2);
          }
        } else {
          for (uint32_t w = 0; w < nw; w++) dst[w] = 0;
          cst = (uint32_t)ETLG_CELL_MISSING;
          flags |= ETLG_FLAG_PARTIAL;
        }
        if (cst) sttab[(SW + (k >> 4)) * cf + lane] |= cst << (2 * (k & 15u));
      }
    }
    // row state words, Begin / Commit bodies, event headers
    if (emit) {
      const uint32_t tag = v.tag;
      if (isrow || tag == 'B' || tag == 'C') {
        const u8* b = v.fr + kBodyOff;
        uint32_t table = rel_id, slot_id = 0;
        uint64_t commit_lsn = tx.final_lsn;
        if (isrow) {
          const DevSlot& s = p.slots[slot];
          slot_id = s.host_id;
          // the 2-bit states at the head of each row image: st_key / st_full bytes (4 per 16 columns)
          const uint32_t so = old_kind == ETLG_OLD_NONE ? 0u : old_kind == ETLG_OLD_KEY ? s.st_key : s.st_full;
          for (uint32_t w = 0; 4 * w < so && w < SW; w++) ((uint32_t*)body)[w] = sttab[w * cf + lane];
          if (tag != 'D') for (uint32_t w = 0; 4 * w < s.st_full && w < SW; w++) ((uint32_t*)(body + old_sz))[w] = sttab[(SW + w) * cf + lane];
        } else if (tag == 'B') {   // parse_event_from_begin_message, codec/event.rs:303-316
          commit_lsn = ld_be64(b); table = ld_be32(b + 16);
          st64((uint32_t*)body, ld_be64(b + 8));
        } else {                   // parse_event_from_commit_message, codec/event.rs:322-336
          flags = b[0]; commit_lsn = ld_be64(b + 1); table = 0;
          st64((uint32_t*)body, ld_be64(b + 9)); st64((uint32_t*)body + 2, ld_be64(b + 17));
        }
        pg.ev_kind[ev_idx] = (u8)tag;
        pg.ev_flags[ev_idx] = (u8)flags;
        pg.ev_table[ev_idx] = table;
        pg.ev_slot[ev_idx] = slot_id;
        pg.ev_start[ev_idx] = ld_be64(v.fr + 6);
        pg.ev_commit[ev_idx] = commit_lsn;
        pg.ev_ord[ev_idx] = tx.ord;
        pg.ev_body[ev_idx] = fx_off;
      } else {
        // Truncate / Relation events (rare): the generic writer
        RowMsg dummy{};
        write_frame(p, v, tx, dummy, -1, ev_idx, fx_off, 0, nullptr, true);
      }
    }
    ETLG_WAVE_PRIO(0);
  }
  TSTAMP(10);
  {
    // ================= C (all waves; the spine joins when H is done): text of String / deferred cells -> heap, 16 bytes per lane and step.
    // Work is claimed, not dealt out: a row with little text whole (the first wave to ask gets it, references and all), a long row step by
    // step — so the waves finish together whenever they arrive.
    const bool cclk = (dbg_u & 8) && role == 1 && lane == 0 && (blockIdx.x & 15) == 3;
    const unsigned long long c_t0 = cclk ? clock64() : 0ull;
    for (uint32_t half = 0; half < 2 && heap_fits && fixed_fits; half++) {
      uint64_t m = half ? m_new : m_old;
      while (m) {
        const uint32_t hr = (uint32_t)__builtin_ctzll(m);
        m &= m - 1;
        const uint32_t r = (half ? maxh_old : 0u) + hr;
        // a row with little text (two steps' worth or less) is one wave's from the references to the last byte; a long one is split by steps
        const uint32_t Trow = row_chunks[r];
        const bool solo = Trow <= 128u;
        if (!solo && (uint32_t)__builtin_amdgcn_readfirstlane((int)row_next[r]) >= Trow) continue;   // (every step of it has been claimed; one lane's reading for the whole wave)
        if (solo) {
          uint32_t cl = 0;
          if (lane == 0) cl = atomicAdd(&row_next[r], 1u);
          if (__builtin_amdgcn_readfirstlane((int)cl) != 0) continue;
        }
        const uint32_t w = lane < nt ? htab[r * cf + lane] : 0u;
        const uint32_t kind = w >> 30;
        const bool act = kind == HK_COPY;
        const uint32_t pos = w & 0x1FFFFu;
        const uint32_t len = kind != HK_NONE ? __builtin_bswap32(ldu32(base + pos - 4)) : 0u;
        const uint32_t dst = kind != HK_NONE ? fr_hp[lane] + (((w >> 17) & kRowsHeapMaxDw) << 2) : 0u;
        // one wave per row (the owner of a short row, the claimant of a long row's first step): every heap cell's slot gets its heap
        // reference; numerics / bytea are emitted at their final place
        uint32_t t0 = 0;
        if (!solo) { uint32_t cl = 0; if (lane == 0) cl = atomicAdd(&row_next[r], 64u); t0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cl); }
        if (t0 == 0 && kind != HK_NONE) {
          uint32_t* const slotp = (uint32_t*)(pg.fixed + pre_fx + ((uint32_t)hslot[r * cf + lane] << 2));
          slotp[0] = dst;
          if (kind != HK_COPY) {
            uint32_t tmp[4] = {0, 0, 0, 0}, st = 0, hcur = dst;
            const uint32_t err = decode_text_cell<false>(kind == HK_NUMERIC ? (uint32_t)ETLG_TC_NUMERIC : (uint32_t)ETLG_TC_BYTEA, base + pos, len, tmp, pg.heap, hcur, st, true);
            if (err) record_error(pg, f0 + lane, RK_DECODE, err);
            else slotp[1] = tmp[1];
          }
        }
        const uint32_t clen = act ? len : 0u;
        const uint32_t nch = (clen + 15u) >> 4;
        const uint32_t incl = wave_scan_add(nch);
        const uint32_t T = wave_last(incl);
        const uint32_t excl = incl - nch;
        volatile uint8_t* const marks = own_mark[tid >> 6];
        for (; t0 < T;) {
          const uint32_t x = t0 + lane;
          // The chunk's cell. Cells lie in lane order, so the cell of chunk x is the last one that starts at or before x: every cell whose
          // first chunk falls into this step leaves its lane number at that position of a 64-byte scratch row, the cell still going when
          // the step begins comes in at position 0, and a running maximum over the positions (DPP) spreads the owners. One LDS round trip
          // and a scan instead of a six-step binary search through ds_bpermute (six dependent round trips).
          marks[lane] = 0;
          ETLG_WAVE_JOIN();
          const uint32_t s_i = excl - t0;
          if (nch != 0 && s_i < 64u) marks[s_i] = (uint8_t)(lane + 1u);
          const unsigned long long before = __ballot(nch != 0 && excl < t0);
          const uint32_t carry = before ? 64u - (uint32_t)__builtin_clzll(before) : 0u;
          uint32_t mk = marks[lane];
          if (lane == 0 && mk == 0) mk = carry;
          const uint32_t own1 = wave_scan_max(mk);
          const uint32_t owner = own1 ? own1 - 1u : 0u;
          const uint32_t o_pos = (uint32_t)__shfl((int)pos, (int)owner, 64), o_len = (uint32_t)__shfl((int)clen, (int)owner, 64);
          const uint32_t o_dst = (uint32_t)__shfl((int)dst, (int)owner, 64), o_excl = (uint32_t)__shfl((int)excl, (int)owner, 64);
          if (x < T) {
            const uint32_t boff = (x - o_excl) << 4;
            const uint32_t rem = o_len - boff;                 // >= 1
            const u8* src = base + o_pos + boff;
            const uint32_t sh = (uint32_t)(uintptr_t)src & 3u;
            const uint32_t* qd = (const uint32_t*)(src - sh);
            const uint32_t need = sh + (rem < 16u ? rem : 16u);   // bytes from qd[0] on
            uint32_t wv[5];
#pragma unroll
            for (uint32_t j = 0; j < 5; j++) wv[j] = 4 * j < need ? qd[j] : 0u;
            uint32_t prev = boff ? __builtin_amdgcn_alignbyte(qd[0], qd[-1], sh) : 0u;
            uint32_t* out = (uint32_t*)(pg.heap + o_dst + boff);
            bool bad = false;
            uint32_t xs[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
              if (4 * j < rem) {
                const uint32_t rj = rem - 4 * j;
                uint32_t xw = __builtin_amdgcn_alignbyte(wv[j + 1], wv[j], sh);
                if (rj < 4) xw &= (1u << (8 * rj)) - 1u;
                xs[j] = xw;
                if ((xw | prev) & 0x80808080u) bad |= utf8_dword_bad(prev, xw, rj == 4);
                prev = xw;
              }
            }
            if (rem > 12u) __builtin_memcpy(out, xs, 16);   // all four dwords are the cell's (the last one zero padded): one 16-byte store (4-byte aligned)
            else {
#pragma unroll
              for (uint32_t j = 0; j < 3; j++) if (4 * j < rem) out[j] = xs[j];
            }
            if (bad) record_error(pg, f0 + owner, RK_DECODE, ETLG_E_UTF8);
          }
          if (solo) t0 += 64u;
          else { uint32_t cl = 0; if (lane == 0) cl = atomicAdd(&row_next[r], 64u); t0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cl); }
        }
      }
    }
    if (cclk) atomicAdd(&pg.res->dbg_t[8], clock64() - c_t0);   // (phase clocks: the text copy as one of its waves sees it)
  }
  TSTAMP(6);
}

}  // namespace etlg

extern "C" {

using namespace etlg;

static int rows_nw() {   // waves per tile: ETLG_ROWS_NW = 2 | 4 (experiments; the shipped value is RNW)
  static const int nw = [] { const char* e = getenv("ETLG_ROWS_NW"); const int v = e ? atoi(e) : RNW; return v == 2 ? 2 : 4; }();
  return nw;
}

void etlg_k_launch_rows(const DecParams* p, const void* qv, hipStream_t s) {
  const FusedParams* q = (const FusedParams*)qv;
  if (rows_nw() == 2) hipLaunchKernelGGL((k_rows<2>), dim3(q->ntiles), dim3(2 * 64), q->lds_bytes, s, *p, *q);
  else hipLaunchKernelGGL((k_rows<4>), dim3(q->ntiles), dim3(4 * 64), q->lds_bytes, s, *p, *q);
}

int etlg_k_rows_set_lds(void) {
  const int a = hipFuncSetAttribute((const void*)k_rows<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 6144) == hipSuccess ? 0 : 1;
  const int b = hipFuncSetAttribute((const void*)k_rows<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 6144) == hipSuccess ? 0 : 1;
  return a | b;
}
int etlg_k_rows_waves(void) { return rows_nw(); }
int etlg_k_rows_waves_per_simd(void) { return ETLG_ROWS_MINBLOCKS; }

// heap-cell table + cell positions next to the image and the window
uint32_t etlg_k_rows_table_bytes(uint32_t maxh_old, uint32_t maxh, uint32_t maxc, uint32_t cf) {
  return (((maxh_old + maxh) * cf * 6u + 15u) & ~15u) + ((2u * ((maxc + 15u) >> 4) * cf * 4u + 15u) & ~15u) + ((((maxc + 31u) >> 5) * cf * 4u + 15u) & ~15u) +
         ((2u * maxc * cf * 2u + 15u) & ~15u) + (4u * maxc + 32u) * 16u;
}
uint32_t etlg_k_rows_static_lds(void) { return 3840u; }   // the kernel's __shared__ arrays + slack
int etlg_k_rows_occupancy(uint32_t lds_bytes) {   // workgroups of k_rows that fit a CU with that much dynamic LDS (debugging aid)
  int n = -1;
  const hipError_t e = rows_nw() == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_rows<2>, 2 * 64, lds_bytes) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_rows<4>, 4 * 64, lds_bytes);
  if (e != hipSuccess) return -1;
  return n;
}
uint32_t etlg_k_rows_max_cols(void) { return 128u; }      // (a task names its cell index in 8 bits, a cell position its frame offset in 16)
uint32_t etlg_k_rows_max_heap_cols(void) { return 64u; }  // rows per image of the heap-cell table (one 64-bit presence mask per image)

}  // extern "C"
