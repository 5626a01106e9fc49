// Column-parallel single-pass decode kernel for gfx950 — the variable-length path.
//
// k_fused (fused.hip) gives every frame one lane; with wide rows (hundreds of bytes of
// TEXT / NUMERIC per frame) that leaves one wave per ~25 KB of staged LDS, i.e. 4 waves
// per CU, and every per-lane character loop is latency bound. Here a tile is 64 frames
// and NW waves; after one wave has sliced the tuples into a cell table in LDS, the
// waves work on different COLUMNS of the same 64 frames:
//
//   P0  stage the tile's bytes + side tables into LDS (all waves, coalesced 16-B loads)
//   P1  wave 0, lane = frame: envelope, tag, tuple walk; cell (offset, length, kind)
//       records go to LDS as table[virtual column][frame]; transaction scan;
//       ownership + schema slot per frame
//   P2  waves over virtual columns (old/key image cells, then new image cells),
//       lane = frame: exact heap bytes per cell (numeric digit groups, deferred text, ...)
//   P2b wave 0: tuple-shape checks, per-frame exclusive prefix over its cells' heap bytes,
//       frame sizes -> wave scan -> two-level look-back (three prefixes on three waves)
//   P3  waves over virtual columns, lane = frame: decode the cell, write its slot and heap
//       entry at the final position; 2-bit states are OR-ed into an LDS word per row
//   P4  wave 0: toast aliasing, row state words, event headers, first error per frame
//
// A wave therefore sees ONE column class at a time (scalar dispatch through a waterfall
// over the distinct classes), all rows of a tile progress in parallel across columns, and
// occupancy is NW waves per tile instead of one. A tile whose bytes do not fit the LDS window
// reads the input in place. Schemas wider than MAXC columns and every error fall back to the
// multi-pass kernels (kernels.hip).
//
// Table-copy rows (etlg_copy_decode) run through the same tile: k_cells<.., COPYK> stages 64 COPY text
// rows instead of 64 frames and cells_tile<.., 2, ..> replaces P1's tuple walk by the COPY field splitter
// (window bitmaps -> row walk in registers -> the fields with backslashes finished by all waves).
#define ETLG_FLOAT_CALL static __device__ __attribute__((noinline))
#define ETLG_DBG_WORD dbg_u   // cells_tile / k_cells keep the debug word in a scalar register of its own
#define ETLG_TSTAMP_WHO (wave == 0 && lane == 0)   // phase clocks are taken by the tile's spine wave (its role rotates, see cells_tile)
#include <type_traits>
#include "lookback.hip.h"
#include "utf8_swar.h"

namespace etlg {

constexpr int CF = 64;         // frames per tile
// Replicated columns per slot: two instantiations. NARROW: <= 16 columns — one 32-bit state word per row image, 16 + 16-bit column
// masks in one register (the tuned cfg3 path; its code is what it was before WIDE existed). WIDE: <= 32 columns — two state words,
// 32 + 32-bit masks in a 64-bit value, column masks derived from the DevCol records on the device (DevSlot keeps its 16-bit fields).
// The host picks by the widest slot a frame of the batch can decode against: a stream whose tables outgrow 16 columns (cfg5's
// ALTER TABLE ADD COLUMN) keeps this kernel instead of falling to k_fused/64.
constexpr int MAXC_NARROW = 16, MAXC_WIDE = 32;
// virtual columns of a tile whose widest slot has maxc columns: [0, maxc) old / key image, [maxc, 2 maxc) new image

enum : uint32_t { CT_N = 0, CT_U = 1, CT_T = 2, CT_B = 3 };  // cell kind
// classes whose heap bytes depend on the text, not only on its length (cell_heap_bytes): float (a deferred text), numeric, bytea
constexpr uint32_t kScanClasses = (1u << ETLG_TC_F32) | (1u << ETLG_TC_F64) | (1u << ETLG_TC_NUMERIC) | (1u << ETLG_TC_BYTEA);

// The cell table of a tile, [virtual column][frame], written by the tuple walk (P1), sized by P2 / P2b, read by P3.
// STAGED tiles keep ONE dword per cell — where the cell's tag byte sits in the LDS window (17 bits) and the cell's heap
// bytes, later its heap offset inside the frame, in dwords (15 bits) — and read kind and length back from the staged bytes:
// at 12 bytes per cell the table of a 12-column schema was 18 KB of a 47 KB tile and capped a CU at three workgroups.
// A tile read in place has no window; it keeps (32-bit offset, length | kind << 30, heap bytes) per cell in the space the
// window would have taken.
// The sizing pass (P2) has to parse a float cell to know whether it is handed back DEFERRED (a heap entry) or not; P3 used to parse it
// again (~37 k cycles per column and tile on the table-copy bench rows, profiles/r03ad_copy_direct.txt). The first kFloatCache
// scan columns of a tile keep their parsed values in LDS instead: [scan rank][frame] bits + one ballot of the lanes that hold a value.
constexpr uint32_t kFloatCache = 4;
constexpr uint32_t kWinMax = 120u * 1024u;   // staged bytes of a tile: tag positions fit 17 bits, a frame's heap bytes fit 15 bits of dwords
// A tile of table-copy rows (k_copy_cells) has no cell headers in its window: (window offset 17 bits | heap dwords 15 bits,
// length | kind << 30) per cell.
template <int TAB> struct CellTab;   // 0: tile read in place, 1: staged WAL tile, 2: staged table-copy tile
template <> struct CellTab<1> {
  uint32_t* e;
  // the heap field starts out as what a class that copies its text takes (String, wholesale-deferred): pad4(len)
  DEV void put(uint32_t i, uint32_t tagpos, uint32_t len, uint32_t kind) const { e[i] = (tagpos << 15) | (kind == CT_T ? (len + 3u) >> 2 : 0u); }
  // -> offset of the cell's text, its length and kind; returns the raw entry (heap field)
  DEV uint32_t get(uint32_t i, const u8* base, uint32_t& pos, uint32_t& len, uint32_t& kind) const {
    const uint32_t w = e[i], c = w >> 15;
    const uint64_t head = ldu64(base + c);  // the window has 16 spare bytes past any frame
    const uint32_t t = (uint32_t)head & 0xFFu;
    const bool is_val = (t == 't') | (t == 'b');
    len = is_val ? __builtin_bswap32((uint32_t)(head >> 8)) : 0u;
    kind = t == 't' ? (uint32_t)CT_T : t == 'b' ? (uint32_t)CT_B : t == 'u' ? (uint32_t)CT_U : (uint32_t)CT_N;
    pos = c + 5;
    return w;
  }
  DEV uint32_t heap_of(uint32_t w) const { return (w & 0x7FFFu) << 2; }
  DEV uint32_t raw(uint32_t i) const { return e[i]; }
  DEV void set_heap_of(uint32_t i, uint32_t w, uint32_t h) const { e[i] = (w & ~0x7FFFu) | (h >> 2); }  // w: what get() / raw() returned; h: a multiple of 4 below 128 KiB
};
template <> struct CellTab<0> {
  uint2* pl; uint32_t* h;
  DEV void put(uint32_t i, uint32_t tagpos, uint32_t len, uint32_t kind) const { pl[i] = make_uint2(tagpos + 5, len | (kind << 30)); h[i] = kind == CT_T ? pad4(len) : 0u; }
  DEV uint32_t get(uint32_t i, const u8*, uint32_t& pos, uint32_t& len, uint32_t& kind) const {
    const uint2 v = pl[i];
    pos = v.x; len = v.y & 0x3FFFFFFFu; kind = v.y >> 30;
    return h[i];
  }
  DEV uint32_t heap_of(uint32_t w) const { return w; }
  DEV uint32_t raw(uint32_t i) const { return h[i]; }
  DEV void set_heap_of(uint32_t i, uint32_t, uint32_t v) const { h[i] = v; }
};
template <> struct CellTab<2> {
  uint2* e;
  DEV void put(uint32_t i, uint32_t pos, uint32_t len, uint32_t kind) const { e[i] = make_uint2(pos | ((kind == CT_T ? (len + 3u) >> 2 : 0u) << 17), len | (kind << 30)); }
  DEV uint32_t get(uint32_t i, const u8*, uint32_t& pos, uint32_t& len, uint32_t& kind) const {
    const uint2 v = e[i];
    pos = v.x & 0x1FFFFu; len = v.y & 0x3FFFFFFFu; kind = v.y >> 30;
    return v.x;
  }
  DEV uint32_t heap_of(uint32_t w) const { return (w >> 17) << 2; }
  DEV uint32_t raw(uint32_t i) const { return e[i].x; }
  DEV void set_heap_of(uint32_t i, uint32_t w, uint32_t h) const { e[i].x = (w & 0x1FFFFu) | ((h >> 2) << 17); }
};

// frame meta word: tag (8) | old_kind (2) << 8 | wire_ok << 10 | emit << 11 | too_wide << 12
DEV uint32_t meta_tag(uint32_t m) { return m & 0xFF; }
DEV uint32_t meta_old(uint32_t m) { return (m >> 8) & 3; }

// Column of the slot that virtual-column cell k of an image with `n` cells decodes against.
// mode: ROW_FULL / ROW_UPDATE -> k; ROW_KEY dense -> k-th identity column; ROW_KEY
// full-width -> k if it is an identity column. Returns -1 when the cell is skipped.
DEV int cell_column(const DevSlot& s, const DevCol* cols, uint32_t mode, uint32_t n, uint32_t k) {
  if (mode != ROW_KEY) return k < s.n_cols ? (int)k : -1;
  if (n == s.n_ident) {
    for (uint32_t c = 0; c < s.n_cols; c++) if (cols[c].identity && cols[c].key_index == k) return (int)c;
    return -1;
  }
  return (k < s.n_cols && cols[k].identity) ? (int)k : -1;
}

// Is image `img` of this frame well shaped against its slot (tuple-level checks of
// convert_tuple_to_row / normalize_key_tuple_to_row, codec/event.rs:559-565, 889-922)?
DEV uint32_t image_shape_error(const DevSlot& s, uint32_t mode, uint32_t n) {
  if (mode == ROW_KEY) {
    if (s.n_ident == 0) return ETLG_E_KEY_MISSING_COLS;
    if (n != s.n_ident && n != s.n_cols) return ETLG_E_KEY_SHAPE;
    return 0;
  }
  return n != s.n_cols ? (uint32_t)ETLG_E_TUPLE_WIDTH : 0u;
}

// Copies the text of up to 64 cells (lane = frame: source offset `pos`, byte length `clen`, heap
// offset `hcur`; clen = 0 for lanes without a cell) into the heap, zero padded to 4 bytes, and
// validates it as UTF-8 on the way. Groups of 2^lgG lanes take one frame at a time, so a frame's
// dwords leave as one contiguous store. Returns (to the frame's own lane) whether its text is
// NOT valid UTF-8. over: the source may be read up to 3 bytes past a cell's end.
DEV bool coop_copy(const u8* base, u8* heap, uint32_t pos, uint32_t clen, uint32_t hcur, uint32_t lane, uint32_t lgG, bool over, uint32_t abl = 0) {
  const uint32_t G = 1u << lgG, grp = lane >> lgG, gl = lane & (G - 1);
  const unsigned long long gmask = ((1ull << G) - 1) << (grp << lgG);  // G <= 32
  bool mine = false;
  // one dword of one frame: load, store, validate
  auto step = [&](const u8* src, uint32_t* dst, uint32_t lj, uint32_t w) -> bool {
    const uint32_t rem = lj - 4 * w;  // >= 1
    uint32_t x = 0, prev = 0;
    if (rem >= 4 || over) {
      __builtin_memcpy(&x, src + 4 * w, 4);
      if (rem < 4) x &= (1u << (8 * rem)) - 1u;
    } else {
      for (uint32_t b2 = 0; b2 < rem; b2++) x |= (uint32_t)src[4 * w + b2] << (8 * b2);
    }
    if (w) __builtin_memcpy(&prev, src + 4 * w - 4, 4);
    if (!(abl & 2)) dst[w] = x;
    if (abl & 1) return x == 0x12345678u;
    return ((x | prev) & 0x80808080u) ? utf8_dword_bad(prev, x, rem == 4) : false;
  };
  // four frames per trip: their shuffles, LDS reads and stores are independent and overlap
  for (uint32_t i = 0; i < G; i += 4) {
    const int j = (int)((grp << lgG) + i);
    uint32_t pj[4], lj[4], hj[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { pj[u] = __shfl(pos, j + u, 64); lj[u] = __shfl(clen, j + u, 64); hj[u] = __shfl(hcur, j + u, 64); }
    bool bad[4];
#pragma unroll
    for (int u = 0; u < 4; u++) bad[u] = gl < ((lj[u] + 3) >> 2) ? step(base + pj[u], (uint32_t*)(heap + hj[u]), lj[u], gl) : false;
#pragma unroll
    for (int u = 0; u < 4; u++) {  // texts longer than one group pass (rare when G fits the column)
      const uint32_t ndw = (lj[u] + 3) >> 2;
      for (uint32_t w = gl + G; w < ndw; w += G) bad[u] |= step(base + pj[u], (uint32_t*)(heap + hj[u]), lj[u], w);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const unsigned long long bal = __ballot(bad[u]);
      if (gl == i + u) mine = (bal & gmask) != 0;
    }
  }
  return mine;
}

// LDS arrays of one tile (declared by the kernel, shared by both instantiations below).
struct CellsLds {
  uint32_t* s_offs;             // CF + 1 frame offsets
  int32_t* fr_slot;             // schema slot of the frame's table, -1 = nothing to decode
  uint32_t* fr_meta;
  uint32_t* fr_n;               // n_old | n_new << 16
  uint64_t* fr_fx;              // fixed-arena offset of the frame's body
  uint32_t* fr_hp;              // heap offset of the frame's first entry
  uint32_t* fr_ev;              // index of the frame's event
  uint32_t (*fr_st)[CF];        // 2-bit cell states: word (img * SW + w) of the old / new row (SW = 1, or 2 in the WIDE kernel)
  uint32_t* fr_err;             // min over (order << 8 | code)
  uint32_t* fr_toast;           // new-row columns sent as 'u'
  uint32_t* s32; uint64_t* s64;
  uint32_t* ct;                 // cell table region (CellTab)
  uint8_t (*vlist)[64];         // virtual columns P2 ([0]) and P3 ([1]) visit
  const uint32_t* bm_sep; const uint32_t* bm_bs; const uint32_t* bm_nl;   // table-copy tiles, one bit per window byte: unescaped tabs / newlines, backslashes, newlines
  uint32_t* cxm;                // table-copy tiles: the columns of each row whose field holds a backslash
  uint64_t (*fcache)[CF]; uint64_t* fc_ok; uint8_t* vinv;   // float4 / float8 cells parsed by the sizing pass, for P3 (see kFloatCache)   // table-copy tiles: where the window's tabs / newlines / backslashes are (one bit per byte)
};

// Everything after staging. STAGED: `base` is the LDS window holding input bytes [b0, ...), reads
// may run up to 15 bytes past a frame; otherwise `base` is the input itself (b0 = 0).
// `p`: parameters whose side-table pointers point at the LDS copy; `pg`: the original ones.
template <int NW, int TAB, bool WIDE>
DEV void cells_tile(const DecParams& p, const DecParams& pg, const FusedParams& q, const CellsLds& sh, const u8* base,
                    uint32_t b0, uint32_t tile, uint32_t nt, uint32_t copy_bad = 0) {
  constexpr bool STAGED = TAB != 0;
  constexpr bool COPY = TAB == 2;   // table-copy rows: the window holds COPY text rows, P1 is the field splitter (copy_walk below)
  uint32_t* const s_offs = sh.s_offs; int32_t* const fr_slot = sh.fr_slot; uint32_t* const fr_meta = sh.fr_meta;
  uint32_t* const fr_n = sh.fr_n; uint64_t* const fr_fx = sh.fr_fx; uint32_t* const fr_hp = sh.fr_hp; uint32_t* const fr_ev = sh.fr_ev;
  uint32_t (*const fr_st)[CF] = sh.fr_st; uint32_t* const fr_err = sh.fr_err; uint32_t* const fr_toast = sh.fr_toast;
  uint32_t* const s32 = sh.s32; uint64_t* const s64 = sh.s64;
  uint8_t (*const vlist)[64] = sh.vlist;
  uint64_t (*const fcache)[CF] = sh.fcache; uint64_t* const fc_ok = sh.fc_ok; uint8_t* const vinv = sh.vinv;
  constexpr uint32_t FC = COPY ? kFloatCache : 0u;   // the float cache exists in table-copy tiles only (WAL tiles keep their LDS for the window)
  constexpr int MAXC = WIDE ? MAXC_WIDE : MAXC_NARROW;   // bits per image in the column masks
  constexpr int SW = WIDE ? 2 : 1;                       // state words per row image
  using mask_t = typename std::conditional<WIDE, uint64_t, uint32_t>::type;
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  // `wave` is a ROLE, not a position: role 0 is the tile's spine (P1, P2b, a look-back, P4 — about twice the instructions of
  // the other roles). The dispatcher places wave w of every 256-thread workgroup on SIMD w of its CU
  // (tools/ubench/simd_place.hip), so with role = position the spine waves of all four resident tiles shared SIMD 0 while
  // the other three SIMDs idled through the single-wave phases. The role rotates with the tile index instead.
  // The launch parameters the phases keep asking for, each in a scalar register of its own: as members of q they sit in an
  // eight-register tuple that is spilled, and every use reloaded all eight (v_readlane x 8, ~250 of them in P3 alone).
  uint32_t maxc, dbg_u, seq_lb;
  ETLG_SCALAR_COPY(maxc, q.maxc); ETLG_SCALAR_COPY(dbg_u, q.dbg); ETLG_SCALAR_COPY(seq_lb, q.seq_lookback);
  const int wave = (int)(((tid >> 6) + ((dbg_u & 0x10000u) ? 0u : tile)) & (uint32_t)(NW - 1));
  const uint32_t VC = 2 * maxc;
  const uint32_t f0 = tile * CF;
  constexpr bool use_lds = STAGED;
  CellTab<TAB> tab;
  if constexpr (COPY) tab.e = (uint2*)sh.ct;
  else if constexpr (STAGED) tab.e = sh.ct;
  else { tab.pl = (uint2*)sh.ct; tab.h = sh.ct + 2 * VC * CF; }
  uint32_t* fail = &pg.res->fused_fail;

  // ================= P1 (wave 0): slice frames into cells, transaction scan, slots
  const bool live = wave == 0 && lane < nt;
  const uint32_t f = f0 + lane;
  FrameView v{f, 0, base, base};
  uint32_t rel_id = 0, old_kind = ETLG_OLD_NONE, n_old = 0, n_new = 0, vbytes = 0, o0 = 0;
  bool wire_ok = true, too_wide = false;
  uint32_t cnt = 0, mark = 0, seg_in = 0, pm = 0, tot_cnt = 0, tot_mark = 0;
  int my_slot = -1;
  mask_t heap_cols = 0;  // virtual columns of this frame whose cells reach the heap (wave 0)
  // The single-wave phases (P1, P2b, the look-backs, P4) are the spine of a tile: every other wave of the workgroup waits
  // for them at a barrier, while the cell phases of the other tiles on this SIMD have slack. They run at a raised issue priority.
  if (wave == 0) {
    ETLG_WAVE_PRIO(3);
    if (lane < CF) { for (int w = 0; w < 2 * SW; w++) fr_st[w][lane] = 0; fr_err[lane] = 0xFFFFFFFFu; fr_toast[lane] = 0; fr_slot[lane] = -1; fr_meta[lane] = 0; fr_n[lane] = 0; }
    if constexpr (COPY) {
      // parse_table_row_from_postgres_copy_bytes (codec/table_row.rs:47-254), lane = row: the row's fields are what lies between its
      // separators, and the bitmaps of the window (k_cells<.., COPYK>: unescaped tabs / newlines, backslashes, newlines) are read
      // eight words at a time into registers, so the walk from separator to separator is register arithmetic (ctz, masks) without
      // a memory round trip per field. Every field gets its cell record (raw text); the fields that hold a backslash are noted
      // in a column mask and finished by all waves afterwards (copy_fix below: the NULL marker, the escapes). Anything else than
      // ncols fields closed by tabs and one final newline — a stray newline, more / fewer columns, a missing terminator, a dangling
      // backslash (it escapes the newline), and for the whole tile invalid UTF-8 (copy_bad) — marks the row as a wire error: the
      // batch then fails and the host decodes it again through the row -> frame rewrite (copy.hip), which knows the reference's
      // error order.
      uint32_t bad = copy_bad, cxm = 0;
      if (live) {
        o0 = s_offs[lane];
        const uint32_t n = s_offs[lane + 1] - o0;   // (the window vote checked the offsets)
        const uint32_t ro = o0 - b0;
        v.fr = base + ro; v.e = base + ro + n; v.tag = 'I';
        cnt = 1;
        rel_id = q.copy_rel;
        const uint32_t ncols = p.slots[pg.copy_slot].n_cols;
        n_new = ncols;
        bad |= n == 0 ? 1u : 0u;
        if (!bad) {
          const uint32_t* const bm_sep = sh.bm_sep; const uint32_t* const bm_bs = sh.bm_bs; const uint32_t* const bm_nl = sh.bm_nl;
          const uint32_t endb = ro + n - 1;   // the row's last byte
          const uint32_t w0 = ro >> 5, w1 = endb >> 5;
          uint32_t k = 0, fs = ro, carry = 0;
          for (uint32_t ws = w0; ws <= w1; ws += 8) {
            uint32_t ms[8], bsw[8], nlw[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) {
              const uint32_t w = ws + u < w1 ? ws + u : w1;   // (words past the row are read as its last word and masked off below)
              ms[u] = bm_sep[w]; bsw[u] = bm_bs[w]; nlw[u] = bm_nl[w];
            }
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) {
              const uint32_t w = ws + u;
              uint32_t rng = w > w1 ? 0u : ~0u;
              rng &= w == w0 ? ~0u << (ro & 31u) : ~0u;
              rng &= w == w1 ? ~0u >> (31u - (endb & 31u)) : ~0u;
              uint32_t m = ms[u] & rng;
              const uint32_t b_ = bsw[u] & rng;
              bad |= (m & nlw[u]) != (w == w1 ? 1u << (endb & 31u) : 0u) ? 1u : 0u;   // the only separator that is a newline is the row's last byte
              uint32_t lo = 0;   // first bit of this word that belongs to the current field
              while (m) {
                const uint32_t bit = (uint32_t)__builtin_ctz(m);
                m &= m - 1u;
                const uint32_t below = (bit ? ~0u >> (32u - bit) : 0u) & (~0u << lo);   // lo <= bit <= 31
                const uint32_t pos = (w << 5) + bit;
                if (k < maxc) tab.put((maxc + k) * CF + lane, fs, pos - fs, (uint32_t)CT_T);
                cxm |= (carry | (b_ & below)) ? 1u << (k & 31u) : 0u;
                vbytes += pos - fs;
                k++; fs = pos + 1u; carry = 0; lo = bit + 1u;
              }
              carry |= lo < 32u ? b_ & (~0u << lo) : 0u;
            }
          }
          bad |= (k != ncols || fs != ro + n || k > maxc) ? 1u : 0u;
        }
        wire_ok = bad == 0;
      }
      if (lane < CF) sh.cxm[lane] = (live && !bad) ? cxm : 0u;
      TSTAMP(11);
    } else {
      if (live) {
        o0 = s_offs[lane];
        const uint32_t o1 = s_offs[lane + 1];
        if (o1 > o0 && o1 <= pg.in_len) {
          v.fr = base + (o0 - b0);
          v.e = base + (o1 - b0);
          v.tag = classify_ptr(v.fr, o1 - o0);
        }
        const uint32_t tag = v.tag;
        if (!(tag == 'I' || tag == 'U' || tag == 'D')) {
          RowMsg dummy;
          wire_ok = frame_structure(v, dummy);
        }
        if (consumes_ordinal(tag)) cnt = 1;
        if (tag == 'B') { cnt |= 0x80000000u; mark = ((o0 + 1) << 1) | 1; }
        if (tag == 'C') mark = (o0 + 1) << 1;
      }
      TSTAMP(11);
      // parse_row_msg + walk_tuple for the I / U / D frames, recording every cell, with 32-bit offsets into `base`.
      // This wave runs alone while the others wait and a lone wave issues an instruction every 5-8 cycles, so the
      // walk is priced per instruction: a frame is at most (image header, cells) twice, and the two kinds of item
      // get their own wave-level steps — one header step per pass, then a cell loop whose body is a dozen VALU
      // operations (no data-dependent branches: bitwise predicates and selects).
      {
        // Loop-carried state is kept in integers (st: 0 idle, 1 in front of an image header, 2 inside an image; bad / wide:
        // error flags): as `bool`s they live in lane-mask SGPR pairs and every trip paid ~30 scalar instructions merging them.
        const uint32_t tag = v.tag;
        const uint32_t upd = tag == 'U' ? 1u : 0u;
        const uint32_t maxc_u = maxc;
        uint32_t c = 0, e = 0, img = tag == 'I' ? 1u : 0u, k = 0, n = 0, st = 0, bad = 0, wide = 0;
        if (live && (tag == 'I' || tag == 'U' || tag == 'D')) {
          c = (uint32_t)(v.fr - base) + kBodyOff; e = (uint32_t)(v.e - base);
          wire_ok = e >= c + 5;
          if (wire_ok) { rel_id = ld_be32(base + c); c += 4; st = 1; }
        }
        // the next 8 bytes of a lane: item tag + i16 count (header) or item tag + i32 length (cell)
        auto next8 = [&](bool on) -> uint64_t {
          if (STAGED) return ldu64(base + c);  // the window has 16 spare bytes past any frame
          uint64_t head = 0;
          if (on) for (uint32_t i = 0; i < 5 && c + i < e; i++) head |= (uint64_t)base[c + i] << (8 * i);
          return head;
        };
        // One loop over "steps": a header step whenever some lane stands in front of an image header (the first trip for
        // everybody, later the frames that go on from an old / key image to the new one), then a cell step for every lane
        // inside an image. A tile finishes in (cells of its longest frame) + 1 or 2 trips.
        for (;;) {
          const unsigned long long wh = __ballot(st == 1);
          if (!(wh | __ballot(st == 2))) break;
          if (wh) {  // image header: 'K' | 'O' | 'N', i16 column count
            const bool on = st == 1;
            const uint64_t head = next8(on);
            const uint32_t t = (uint32_t)head & 0xFFu;
            const uint32_t room = e - c;  // c <= e holds for a lane that is still going
            const bool is_old = (t == 'K') | (t == 'O');
            const uint32_t img_h = ((img == 0) & !is_old & (upd != 0)) ? 1u : img;  // update without an old image
            const uint32_t cnt16 = (((uint32_t)head >> 8) & 0xFFu) << 8 | (((uint32_t)head >> 16) & 0xFFu);
            const bool hdr_ok = (room >= 3) & (img_h == 0 ? is_old : t == 'N') & !(cnt16 & 0x8000u);
            const bool go = on & hdr_ok;
            bad |= (on & !hdr_ok) ? 1u : 0u;
            wide |= (go & (cnt16 > maxc_u)) ? 1u : 0u;
            c += go ? 3u : 0u;
            const bool first = go & (img_h == 0);
            old_kind = first ? (t == 'K' ? (uint32_t)ETLG_OLD_KEY : (uint32_t)ETLG_OLD_FULL) : old_kind;
            n_old = first ? cnt16 : n_old;
            n_new = (go & (img_h != 0)) ? cnt16 : n_new;
            n = go ? cnt16 : n;
            k = go ? 0u : k;
            const bool again = go & (cnt16 == 0) & (img_h == 0) & (upd != 0);  // an image without cells is complete at once
            img = go ? (again ? 1u : img_h) : img;
            st = on ? (go ? (cnt16 == 0 ? (again ? 1u : 0u) : 2u) : 0u) : st;
            if (!__ballot(st == 2)) continue;
          }
          {  // cells: 'n' | 'u' | ('t' | 'b') i32 len bytes
            const bool on = st == 2;
            const uint64_t head = next8(on);
            const uint32_t t = (uint32_t)head & 0xFFu;
            const uint32_t room = e - c;
            const bool is_val = (t == 't') | (t == 'b');
            const uint32_t len = is_val ? __builtin_bswap32((uint32_t)(head >> 8)) : 0u;
            const uint32_t kind = t == 't' ? (uint32_t)CT_T : t == 'b' ? (uint32_t)CT_B : t == 'u' ? (uint32_t)CT_U : (uint32_t)CT_N;
            const bool cell_ok = (room >= 1) & (is_val | (t == 'n') | (t == 'u')) & (!is_val | ((room >= 5) & (len <= room - 5)));
            const bool go = on & cell_ok;
            if (go & (k < maxc_u)) tab.put((img * maxc_u + k) * CF + lane, c, len, kind);
            bad |= (on & !cell_ok) ? 1u : 0u;
            wide |= (go & (len > 0x3FFFFFFFu)) ? 1u : 0u;
            vbytes += go ? len : 0u;
            c += go ? (is_val ? 5u + len : 1u) : 0u;
            k += go ? 1u : 0u;
            // image complete: an update goes on to its new image (a header step), everything else is finished
            const bool done_img = go & (k == n);
            const bool again = done_img & (img == 0) & (upd != 0);
            img = again ? 1u : img;
            st = on ? (go ? (done_img ? (again ? 1u : 0u) : 2u) : 0u) : st;
          }
        }
        wire_ok = wire_ok & (bad == 0);
        too_wide |= wide != 0;
      }
    }
    TSTAMP(9);
    // wave-level transaction scan (no barrier: a tile's frames live in one wave)
    {
      const uint32_t ic = wave_scan_incl(cnt, [](uint32_t a, uint32_t b) { return seg_combine(a, b); }, 0u);
      const uint32_t im = wave_scan_max(mark);
      pm = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)im, 0x138, 0xF, 0xF, false);  // previous lane's value, 0 into lane 0
      seg_in = ic;
      tot_cnt = wave_last(ic); tot_mark = wave_last(im);
    }
  }
  const uint64_t txn_agg = ((uint64_t)seg_pack30(tot_cnt) << 32) | tot_mark;  // meaningful on wave 0
  TxnCtx tx{true, 0, 0};
  uint32_t bc = 0, bm = 0;
  uint64_t carried_lsn = 0, start_ord = 0;   // wave 0, from the transaction look-back: final_lsn of the Begin `bm` names, the ordinal the batch starts from
  auto make_tx = [&](const TxnStart& t) {
    bc = t.seg; bm = t.mark; carried_lsn = t.lsn; start_ord = t.ord;
    const uint32_t seg = seg_combine(bc, seg_in);
    const uint32_t last = bm > pm ? bm : pm;
    tx.in_txn = (last & 1u) != 0;
    // carried-in Begin: fetched by the wave that ran the transaction look-back; a Begin of this tile: read in place
    tx.final_lsn = !tx.in_txn ? 0 : last == bm ? carried_lsn : ld_be64(base + (((last >> 1) - 1) - b0) + kBodyOff);
    const uint64_t c = seg & 0x7FFFFFFFu;
    tx.ord = (seg & 0x80000000u) ? c - 1 : start_ord + c - 1;
  };
  if (wave == 0) {
    if (seq_lb) {
      make_tx(txn_lookback(pg, q.d_txn, q.ntiles, tile, txn_agg, fail, s32, s64));   // (in registers: no barrier between the look-back and here)
    }
    TSTAMP(10);
    if (live && too_wide) atomicOr(fail, 4u);
    // ownership + schema slot (size_frame's lookups), per frame
    if (live && wire_ok && (v.tag == 'I' || v.tag == 'U' || v.tag == 'D')) {
      const int ti = (pg.flags & 2u) ? -1 : find_table(p, rel_id);
      int slot = -1;
      if (pg.flags & 2u) slot = pg.copy_slot;  // table-copy rows: the caller named the schema
      else if (should_apply(p, ti, rel_id, tx.final_lsn)) {
        slot = cache_slot_before(p, ti, f);
        if (slot < 0) { record_error(pg, f, RK_SCHEMA, (uint32_t)(-slot)); slot = -1; }
        else if (p.slots[slot].n_cols > maxc) { atomicOr(fail, 4u); slot = -1; }
      }
      fr_slot[lane] = slot;
      my_slot = slot;
    }
    if (live) {
      fr_meta[lane] = v.tag | (old_kind << 8) | ((wire_ok ? 1u : 0u) << 10);
      fr_n[lane] = n_old | (n_new << 16);
    }
    // Which virtual columns hold cells of a decodable image at all (P3 visits those), and which of them belong to a class
    // whose heap bytes depend on the text (P2 visits those; every other cell keeps the walk's pad4(len), counted in P2b
    // only where the column's class reaches the heap). Bits 0..15: old / key image, 16..31: new image.
    mask_t present = 0, scan = 0;
    const mask_t one = 1;
    if (my_slot >= 0) {
      const DevSlot& s = p.slots[my_slot];
      const uint32_t tag = v.tag;
      const mask_t po = old_kind != ETLG_OLD_NONE ? (one << (n_old < maxc ? n_old : maxc)) - one : (mask_t)0;
      const mask_t pn = tag != 'D' ? (one << (n_new < maxc ? n_new : maxc)) - one : (mask_t)0;
      mask_t fh, fs, oh, os;   // heap / scan columns of a full row; of this frame's old image
      if constexpr (!WIDE) {
        const uint32_t full = s.has_var;
        fh = full & 0xFFFFu; fs = full >> 16;
        oh = fh; os = fs;
        if (old_kind == ETLG_OLD_KEY) {
          if (n_old == s.n_ident) { oh = s.key_masks & 0xFFFFu; os = s.key_masks >> 16; }   // dense key tuple: cell j = identity column j
          else { oh &= s.ident_mask; os &= s.ident_mask; }                                   // full-width key tuple: the other cells are skipped unread
        }
      } else {
        // up to 32 columns: the masks come from the column records (DevSlot's fields hold 16 columns)
        const DevCol* cols = p.cols + s.cols_base;
        uint32_t h = 0, sc = 0, id = 0, kh = 0, ks = 0;
        for (uint32_t k = 0; k < s.n_cols; k++) {
          const DevCol cd = cols[k];
          const uint32_t cls = cd.cls;
          const bool heap = !(cls == ETLG_TC_BOOL || cls == ETLG_TC_I16 || cls == ETLG_TC_I32 || cls == ETLG_TC_I64 || cls == ETLG_TC_U32 || cls == ETLG_TC_UUID ||
                              cls == ETLG_TC_DATE || cls == ETLG_TC_TIME || cls == ETLG_TC_TIMETZ || cls == ETLG_TC_TIMESTAMP || cls == ETLG_TC_TIMESTAMPTZ);
          const bool sn = heap && ((kScanClasses >> cls) & 1u);
          h |= heap ? 1u << k : 0u; sc |= sn ? 1u << k : 0u;
          if (cd.identity) { id |= 1u << k; if (cd.key_index < 32) { kh |= heap ? 1u << cd.key_index : 0u; ks |= sn ? 1u << cd.key_index : 0u; } }
        }
        fh = h; fs = sc; oh = h; os = sc;
        if (old_kind == ETLG_OLD_KEY) {
          if (n_old == s.n_ident) { oh = kh; os = ks; } else { oh &= id; os &= id; }
        }
      }
      present = po | (pn << MAXC);
      heap_cols = (oh & po) | ((fh & pn) << MAXC);
      scan = (os & po) | ((fs & pn) << MAXC);
    }
    auto wave_or = [](mask_t m) -> mask_t {
      auto or32 = [](uint32_t x) { return wave_last(wave_scan_incl(x, [](uint32_t a, uint32_t b) { return a | b; }, 0u)); };
      if constexpr (WIDE) return (mask_t)or32((uint32_t)m) | ((mask_t)or32((uint32_t)((uint64_t)m >> 32)) << 32);
      else return (mask_t)or32((uint32_t)m);
    };
    auto pop = [](mask_t m) -> uint32_t { if constexpr (WIDE) return (uint32_t)__builtin_popcountll((uint64_t)m); else return (uint32_t)__builtin_popcount((uint32_t)m); };
    present = wave_or(present);
    scan = wave_or(scan);
    const mask_t heapy = wave_or(heap_cols);
    if (lane < 2u * MAXC) {  // the two visiting lists; P3's starts with the columns that cost most (text to copy, then text to scan),
                            // so that the waves pulling from it finish close to each other
      const mask_t below = (one << lane) - one;
      const uint32_t vc_of = lane < (uint32_t)MAXC ? lane : maxc + (lane - MAXC);
      if ((scan >> lane) & 1u) { const uint32_t rk = pop(scan & below); vlist[0][rk] = (uint8_t)vc_of; if constexpr (FC != 0) vinv[vc_of] = (uint8_t)rk; }
      else if constexpr (FC != 0) vinv[vc_of] = 0xFF;
      const mask_t t1 = present & heapy & ~scan, t2 = present & scan, t3 = present & ~(heapy | scan);
      const uint32_t rank = ((t1 >> lane) & 1u) ? pop(t1 & below)
                          : ((t2 >> lane) & 1u) ? pop(t1) + pop(t2 & below)
                                                : pop(t1 | t2) + pop(t3 & below);
      if ((present >> lane) & 1u) vlist[1][rank] = (uint8_t)vc_of;
    }
    if (lane == 0) { s32[10] = pop(scan); s32[11] = pop(present); }
    ETLG_WAVE_PRIO(0);
  }
  __syncthreads();
  if constexpr (COPY) {
    // copy_fix: the fields that hold a backslash, one per lane (wave = column, lane = row). A field that is exactly `\N` on the raw
    // bytes is NULL (table_row.rs:199); in every other one the escapes are undone IN PLACE (:129-176; unescaping only shrinks, so
    // the text is rewritten from the field's first byte and the cell record keeps its position). A backslash is never the
    // field's last byte: the separator behind it would have been an escaped character (k_cells<.., COPYK>).
    const uint32_t ncols = p.slots[pg.copy_slot].n_cols;
    const uint32_t* const bm_bs = sh.bm_bs;
    u8* const win = const_cast<u8*>(base);
    auto next_bs = [&](uint32_t from, uint32_t lim) -> uint32_t {   // next backslash in [from, lim), or lim
      while (from < lim) {
        const uint32_t m = bm_bs[from >> 5] >> (from & 31u);
        if (m) { const uint32_t q2 = from + (uint32_t)__builtin_ctz(m); return q2 < lim ? q2 : lim; }
        from = (from | 31u) + 1u;
      }
      return lim;
    };
    for (uint32_t k = (uint32_t)wave; k < ncols && k < maxc; k += NW) {
      if (lane < nt && ((sh.cxm[lane] >> k) & 1u)) {
        uint32_t start, raw_len, kind0;
        tab.get((maxc + k) * CF + lane, base, start, raw_len, kind0);
        const uint32_t pk = start + raw_len;
        uint32_t nb = next_bs(start, pk);
        if (raw_len == 2 && nb == start && win[start + 1] == 'N') tab.put((maxc + k) * CF + lane, start, 0u, (uint32_t)CT_N);
        else {
          uint32_t pos = start, w_ = start;
          for (;;) {
            const uint32_t run = nb - pos;
            if (w_ != pos) {
              uint32_t j = 0;
              for (; j + 8 <= run; j += 8) {
                const uint64_t x = ldu64(win + pos + j);
#pragma unroll
                for (uint32_t b2 = 0; b2 < 8; b2++) win[w_ + j + b2] = (u8)(x >> (8 * b2));
              }
              for (; j < run; j++) win[w_ + j] = win[pos + j];
            }
            w_ += run; pos = nb;
            if (pos + 1 >= pk) break;   // the field is done. (A backslash as its LAST byte cannot occur in a row that decodes; it can in a
                                        // tile whose bitmaps were built across a malformed row — a row without its newline that ends in
                                        // backslashes, in front of one that starts with some: the batch fails on that row, and nothing
                                        // here reads or writes outside the field.)
            const uint32_t e = win[pos + 1];
            if (e < 0x80u) {
              const uint32_t ch = e == 'b' ? 8u : e == 'f' ? 12u : e == 'n' ? (uint32_t)'\n' : e == 'r' ? (uint32_t)'\r' : e == 't' ? (uint32_t)'\t' : e == 'v' ? 11u : e;
              win[w_++] = (u8)ch;
              pos += 2;
            } else {  // a whole multi-byte character (the window is valid UTF-8)
              const uint32_t l = e >= 0xF0u ? 4u : e >= 0xE0u ? 3u : 2u;
              for (uint32_t t2 = 0; t2 < l; t2++) win[w_ + t2] = win[pos + 1 + t2];
              w_ += l; pos += 1 + l;
            }
            if (pos > pk) pos = pk;   // (only behind a mis-split field, see above)
            nb = next_bs(pos, pk);
          }
          tab.put((maxc + k) * CF + lane, start, w_ - start, (uint32_t)CT_T);
        }
      }
    }
    __syncthreads();
  }
  TSTAMP(2);

  // ================= P2: heap bytes per cell; waves pull virtual columns from a queue
  const uint32_t n_scan = s32[10], n_present = s32[11];
  for (;;) {
    uint32_t vj = 0;
    if (lane == 0) vj = atomicAdd(&s32[8], 1u);
    vj = __builtin_amdgcn_readfirstlane(vj);
    if (vj >= n_scan) break;
    const uint32_t vc = vlist[0][vj];
    const int slot = fr_slot[lane];
    const uint32_t meta = fr_meta[lane];
    uint64_t fbits = 0; bool fok = false;
    const uint32_t img = vc >= maxc, k = vc - img * maxc;
    const uint32_t n = img ? (fr_n[lane] >> 16) : (fr_n[lane] & 0xFFFF);
    const uint32_t tag = meta_tag(meta);
    if (slot >= 0 && k < n && !((img == 0 && meta_old(meta) == ETLG_OLD_NONE) || (img == 1 && tag == 'D'))) {
      const DevSlot& s = p.slots[slot];
      const uint32_t mode = img ? (tag == 'U' ? (uint32_t)ROW_UPDATE : (uint32_t)ROW_FULL)
                                : (meta_old(meta) == ETLG_OLD_KEY ? (uint32_t)ROW_KEY : (uint32_t)ROW_FULL);
      const DevCol* cols = p.cols + s.cols_base;
      const int ci = (!s.has_var || image_shape_error(s, mode, n)) ? -1 : cell_column(s, cols, mode, n, k);
      uint32_t pos, len, kind;
      const uint32_t ent = tab.get(vc * CF + lane, base, pos, len, kind);
      const uint32_t cls = ci >= 0 ? (uint32_t)cols[ci].cls : 0u;
      if (ci >= 0 && kind == CT_T && ((kScanClasses >> cls) & 1u)) {
        const u8* d = base + pos;
        uint32_t h = 0;
        bool done = false;
        while (!done) {  // one pass per distinct class among the active lanes, scalar dispatch inside
          const uint32_t u = __builtin_amdgcn_readfirstlane(cls);
          if (cls == u) {
            if (FC != 0 && (u == ETLG_TC_F32 || u == ETLG_TC_F64)) {  // parsed once: P3 takes the value from the cache
              const int r = parse_float_fast(d, len, u == ETLG_TC_F32, fbits, use_lds);
              h = r == 1 ? pad4(len) : 0u;
              fok = r == 0;
            } else h = cell_heap_bytes(u, d, len, use_lds);
            done = true;
          }
        }
        tab.set_heap_of(vc * CF + lane, ent, h);
      }
    }
    ETLG_WAVE_JOIN();
    if (FC != 0 && vj < FC) {
      const unsigned long long okm = __ballot(fok);
      if (fok) fcache[vj][lane] = fbits;
      if (lane == 0) fc_ok[vj] = okm;
    }
  }
  __syncthreads();
  TSTAMP(3);
  ETLG_WAVE_PRIO(3);   // P2b on wave 0, then one look-back per wave

  // ================= P2b (wave 0): shapes, per-frame heap prefix, sizes, look-back
  uint32_t emit = 0, fixed = 0, heap = 0, old_sz = 0, x_ev = 0, x_fx = 0, x_hp = 0, cells = 0;
  uint64_t pay[3] = {0, 0, 0};
  int row_slot = -1;
  if (wave == 0) {
    if (live) {
      const uint32_t tag = v.tag;
      if (tag == 'I' || tag == 'U' || tag == 'D') {
        if (!wire_ok) record_error(pg, f, RK_WIRE, ETLG_E_WIRE);
        else {
          if (seq_lb && !tx.in_txn) record_error(pg, f, RK_TXN, ETLG_E_TXN_STATE);
          pay[tag == 'I' ? 0 : tag == 'U' ? 1 : 2] = vbytes;
          row_slot = fr_slot[lane];
          if (row_slot >= 0) {
            const DevSlot& s = p.slots[row_slot];
            emit = 1;
            uint32_t cells_old = 0, cells_new = 0;  // cells of each image that reach the heap walk
            for (uint32_t img = (tag == 'I' || old_kind == ETLG_OLD_NONE) ? 1u : 0u; img < (tag == 'D' ? 1u : 2u); img++) {
              const uint32_t mode = img ? (tag == 'U' ? (uint32_t)ROW_UPDATE : (uint32_t)ROW_FULL)
                                        : (old_kind == ETLG_OLD_KEY ? (uint32_t)ROW_KEY : (uint32_t)ROW_FULL);
              const uint32_t n = img ? n_new : n_old;
              const uint32_t rb = mode == ROW_KEY ? s.row_key : s.row_full;
              fixed += rb;
              if (img == 0) old_sz = rb;
              const uint32_t serr = image_shape_error(s, mode, n);
              const uint32_t nc = n < maxc ? n : maxc;
              if (serr) atomicMin(&fr_err[lane], ((img * 32u) << 8) | serr);
              else if (img) cells_new = nc; else cells_old = nc;
            }
            cells = cells_old | (cells_new << 16);
          }
        }
      } else {
        RowMsg dummy{};
        size_frame(p, v, tx, wire_ok, dummy, emit, fixed, heap, pay, row_slot, seq_lb != 0);
      }
    }
    {  // exclusive prefix of the cells' heap bytes in tuple order (old image, then new image): all
       // loads first (they are independent), then the running sum out of registers. Lanes without
       // a decodable row have cells = 0 and only rewrite entries nobody reads.
      const uint32_t cells_old = cells & 0xFFFFu, cells_new = cells >> 16;
      if constexpr (!WIDE) {
        uint32_t hh[2 * MAXC];
#pragma unroll
        for (uint32_t i = 0; i < 2u * MAXC; i++) hh[i] = i < VC ? tab.raw(i * CF + lane) : 0u;
#pragma unroll
        for (uint32_t i = 0; i < 2u * MAXC; i++) {
          if (i < VC && !(COPY && i < maxc)) {  // uniform (a table-copy tile has no old / key image)
            const bool in_new = i >= maxc;
            const uint32_t k = in_new ? i - maxc : i;
            const bool take = (k < (in_new ? cells_new : cells_old)) & (((heap_cols >> (in_new ? 16u + k : k)) & 1u) != 0);
            tab.set_heap_of(i * CF + lane, hh[i], heap);
            heap += take ? tab.heap_of(hh[i]) : 0u;
          }
        }
      } else {   // 64 entries in registers would cost the kernel its occupancy: four at a time
        for (uint32_t i0 = 0; i0 < VC; i0 += 4) {
          uint32_t hh[4];
#pragma unroll
          for (uint32_t u = 0; u < 4; u++) hh[u] = i0 + u < VC ? tab.raw((i0 + u) * CF + lane) : 0u;
#pragma unroll
          for (uint32_t u = 0; u < 4; u++) {
            const uint32_t i = i0 + u;
            if (i < VC) {
              const bool in_new = i >= maxc;
              const uint32_t k = in_new ? i - maxc : i;
              const bool take = (k < (in_new ? cells_new : cells_old)) & (((heap_cols >> (in_new ? (uint32_t)MAXC + k : k)) & 1u) != 0);
              tab.set_heap_of(i * CF + lane, hh[u], heap);
              heap += take ? tab.heap_of(hh[u]) : 0u;
            }
          }
        }
      }
    }
    // wave scan of (events, fixed dwords, heap dwords)
    const uint32_t ie = wave_scan_add(emit), ifx = wave_scan_add(fixed >> 2), ih = wave_scan_add(heap >> 2);
    const uint32_t tot_e = wave_last(ie), tot_f = wave_last(ifx), tot_h = wave_last(ih);
    // payload counters (a batch is < 2 GiB on this path: 32-bit partial sums)
    const uint32_t a0p = wave_last(wave_scan_add((uint32_t)pay[0])), a1p = wave_last(wave_scan_add((uint32_t)pay[1])),
                   a2p = wave_last(wave_scan_add((uint32_t)pay[2]));
    if (lane == 0) {
      if (a0p) atomicAdd(&pg.res->pay_shard[tile & 31][0], (unsigned long long)a0p);
      if (a1p) atomicAdd(&pg.res->pay_shard[tile & 31][1], (unsigned long long)a1p);
      if (a2p) atomicAdd(&pg.res->pay_shard[tile & 31][2], (unsigned long long)a2p);
      s64[0] = ((uint64_t)tot_e << 32) | tot_h;  // aggregates for the other look-back waves
      s64[1] = tot_f;
      s64[2] = txn_agg;
    }
    x_ev = ie - emit; x_fx = ifx - (fixed >> 2); x_hp = ih - (heap >> 2);  // exclusive, inside the tile
  }
  __syncthreads();
  TSTAMP(4);
#ifdef ETLG_CELLS_NOLB_ABLATION
  // measurement build only (tools/build_variant.py ... -DETLG_CELLS_NOLB_ABLATION, results are wrong; not a runtime bit: a branch here cost
  // the shipped kernel its spill-free register allocation): with ETLG_FUSED_DBG bit 17 no look-back at all — made-up prefixes of a
  // plausible size, no descriptor traffic: what a tile would cost if its prefixes were given (the question the plan kernel's pre-pass
  // answered)
  if (dbg_u & 0x20000u) {
    if (wave == 0 && lane == 0) { s64[4] = ((uint64_t)(tile * 64u) << 32) | (tile * 64u * 48u); s64[5] = (uint64_t)tile * 64u * 40u; s32[12] = 0; s32[13] = 1u; s64[6] = pg.final_lsn; s64[3] = pg.next_ord; }
  } else
#endif
  {
  if (wave == 0) { const uint64_t a = lookback<OpAdd2>(q.d_outa, q.d_outa + q.ntiles, tile, s64[0], 0, fail); if (lane == 0) s64[4] = a; }
  if (wave == 1 % NW && NW > 1) { const uint64_t b = lookback<OpAdd>(q.d_outb, q.d_outb + q.ntiles, tile, s64[1], 0, fail); if (lane == 0) s64[5] = b; }
  if (wave == 2 % NW && NW > 2 && !seq_lb) txn_lookback(pg, q.d_txn, q.ntiles, tile, s64[2], fail, s32, s64);
  }
  if (NW <= 2 && wave == 0) {
    if (NW == 1) { const uint64_t b = lookback<OpAdd>(q.d_outb, q.d_outb + q.ntiles, tile, s64[1], 0, fail); if (lane == 0) s64[5] = b; }
    if (!seq_lb) txn_lookback(pg, q.d_txn, q.ntiles, tile, s64[2], fail, s32, s64);
  }
  __syncthreads();
  TSTAMP(5);
  const uint64_t pre_ev = s64[4] >> 32, pre_hp = (uint64_t)(uint32_t)s64[4] << 2, pre_fx = s64[5] << 2;
  // (event index and arena offsets of the frame go to LDS here and come back in P4: kept in registers across P3 they were
  // the 64-bit values the allocator spilled to scratch)
  if (wave == 0) {
    if (!seq_lb) {
      make_tx(TxnStart{s32[12], s32[13], s64[6], s64[3]});   // written by the look-back wave before the barrier above
      if (live && wire_ok) txn_check_frame(pg, v, tx);
    }
    const uint64_t ev_idx = pre_ev + x_ev;
    const uint64_t fx_off = pre_fx + ((uint64_t)x_fx << 2);
    const uint64_t hp_off = pre_hp + ((uint64_t)x_hp << 2);
    if (lane == 0 && tile == q.ntiles - 1) {
      DevResult* r = pg.res;
      r->n_events = pre_ev + (s64[0] >> 32); r->fixed_bytes = pre_fx + (s64[1] << 2); r->heap_bytes = pre_hp + ((uint64_t)(uint32_t)s64[0] << 2);
      r->n_frames = pg.nframes;
      const uint32_t sg = seg_combine(bc, tot_cnt);
      const uint32_t lm = bm > tot_mark ? bm : tot_mark;
      const bool it = (lm & 1u) != 0;
      r->out_in_txn = it;
      r->out_final_lsn = it ? (lm == bm ? carried_lsn : final_lsn_of_mark(pg, lm)) : 0;
      const uint64_t c = sg & 0x7FFFFFFFu;
      r->out_next_ord = (sg & 0x80000000u) ? c : start_ord + c;
      carry_publish(r);
    }
    if (emit && (fx_off + fixed > pg.fixed_cap || hp_off + heap > pg.heap_cap || hp_off + heap > 0xFFFFFFFFull)) {
      record_error(pg, f, RK_DECODE, ETLG_E_WIRE);
      emit = 0;
    }
    fr_fx[lane] = fx_off;
    fr_hp[lane] = (uint32_t)hp_off;
    fr_ev[lane] = (uint32_t)ev_idx;  // a batch has fewer than 2^32 frames
    if (live) fr_meta[lane] |= (emit ? 1u : 0u) << 11;
  }
  ETLG_WAVE_PRIO(0);
  __syncthreads();
  TSTAMP(6);

  // ================= P3: decode cells; waves pull virtual columns from a queue
  for (;;) {
    uint32_t vj = 0;
    if (lane == 0) vj = atomicAdd(&s32[9], 1u);
    vj = __builtin_amdgcn_readfirstlane(vj);
    if (vj >= n_present) break;
    const uint32_t vc = vlist[1][vj];
    const uint32_t img = vc >= maxc, k = vc - img * maxc;
    const uint32_t meta = fr_meta[lane];
    const int slot = fr_slot[lane];
    const uint32_t n = img ? (fr_n[lane] >> 16) : (fr_n[lane] & 0xFFFF);
    const uint32_t tag = meta_tag(meta), ok = meta_old(meta);
    const uint32_t mode = img ? (tag == 'U' ? (uint32_t)ROW_UPDATE : (uint32_t)ROW_FULL)
                              : (ok == ETLG_OLD_KEY ? (uint32_t)ROW_KEY : (uint32_t)ROW_FULL);
    bool act = ((meta >> 11) & 1) && slot >= 0 && k < n && !((img == 0 && ok == ETLG_OLD_NONE) || (img == 1 && tag == 'D'));
    DevCol col{};
    uint32_t* slotp = nullptr;
    uint32_t kout = 0;
    if (act) {
      const DevSlot& s = p.slots[slot];
      const DevCol* cols = p.cols + s.cols_base;
      const int ci = image_shape_error(s, mode, n) ? -1 : cell_column(s, cols, mode, n, k);
      if (ci < 0) act = false;
      else {
        col = cols[ci];
        const uint32_t osz = ok == ETLG_OLD_FULL ? s.row_full : ok == ETLG_OLD_KEY ? s.row_key : 0;
        u8* row = pg.fixed + fr_fx[lane] + (img ? osz : 0);
        slotp = (uint32_t*)(row + (mode == ROW_KEY ? col.off_key : col.off_full));
        kout = mode == ROW_KEY ? col.key_index : (uint32_t)ci;
      }
    }
    uint32_t kind = CT_N, len = 0, pos = 0, hcur = 0;
    if (act) hcur = fr_hp[lane] + tab.heap_of(tab.get(vc * CF + lane, base, pos, len, kind));
    const uint32_t order = img * 32u + 1u + k;
    const uint32_t cls = col.cls;
    uint32_t st = ETLG_CELL_NULL, err = 0;
    bool textual = act && kind == CT_T;
    if (dbg_u >> 6) {  // profiling ablations (results are wrong): skip one family of value codecs
      const uint32_t fam = cls == ETLG_TC_NUMERIC ? 1u : (cls >= ETLG_TC_DATE && cls <= ETLG_TC_TIMESTAMPTZ) ? 2u : cls == ETLG_TC_UUID ? 4u
                         : (cls == ETLG_TC_STRING || class_always_deferred(cls)) ? 8u : 16u;
      if ((dbg_u >> 6) & fam) textual = false;
    }
    // text that is copied to the heap verbatim (String cells, cells deferred wholesale): the
    // whole wave moves the bytes, a group of lanes per frame, with coalesced dword stores
    const bool coop = textual && (cls == ETLG_TC_STRING || class_always_deferred(cls));
    if (__ballot(coop)) {
      const uint32_t clen = coop ? len : 0u;
      const uint32_t lgG = __ballot(clen > 64u) ? 4u : 3u;  // 16 or 8 lanes per frame; longer texts take further visits of the group. Wider groups (32 lanes
                                                           // above 64 bytes) mean more trips of the wave and measured 2-3 % slower, 8 lanes throughout 4-8 % slower
      const bool bad_utf8 = ((dbg_u >> 11) & 4) ? false : coop_copy(base, pg.heap, pos, clen, hcur, lane, lgG, use_lds, (dbg_u >> 11) & 3);
      if (coop) {
        slotp[0] = hcur; slotp[1] = len;
        st = cls == ETLG_TC_STRING ? (uint32_t)ETLG_CELL_VALUE : (uint32_t)ETLG_CELL_DEFERRED;
        if (bad_utf8) err = ETLG_E_UTF8;
      }
    }
    if (FC != 0 && textual && !coop && (cls == ETLG_TC_F32 || cls == ETLG_TC_F64)) {  // a float the sizing pass has parsed
      const uint32_t rk = vinv[vc];
      if (rk < FC && ((fc_ok[rk] >> lane) & 1ull)) { st64(slotp, fcache[rk][lane]); st = ETLG_CELL_VALUE; textual = false; }
    }
    if (textual && !coop) {
      bool done = false;
      while (!done) {  // waterfall over the distinct classes of this wave's cells
        const uint32_t u = __builtin_amdgcn_readfirstlane(cls);
        if (cls == u) { err = decode_text_cell<false>(u, base + pos, len, slotp, pg.heap, hcur, st, use_lds); done = true; }
      }
    } else if (act && kind == CT_N) {
      if (!col.nullable && !(pg.flags & 2u)) err = ETLG_E_REQUIRED_NULL; else slot_zero(slotp, cls);
    } else if (act && kind == CT_U) {
      if (mode == ROW_FULL) err = ETLG_E_FULL_ROW_MISSING;
      else if (mode == ROW_KEY) err = ETLG_E_KEY_MISSING_VALUE;
      else { atomicOr(&fr_toast[lane], 1u << k); st = 0; }  // resolved by the frame's lane in P4
    } else if (act && kind == CT_B) {
      err = ETLG_E_BINARY_FORMAT;
    }
    ETLG_WAVE_JOIN();
    if (act) {
      if (err) atomicMin(&fr_err[lane], (order << 8) | err);
      else if (st) atomicOr(&fr_st[img * SW + (WIDE ? kout >> 4 : 0u)][lane], st << (2 * (WIDE ? kout & 15u : kout)));
    }
  }
  __threadfence_block();
  __syncthreads();
  TSTAMP(7);

  // ================= P4 (wave 0): finish rows, event headers
  if (wave != 0 || !emit) return;
  if (dbg_u & 2) return;
  ETLG_WAVE_PRIO(3);
  const uint32_t tag = v.tag;
  const uint64_t ev_idx = fr_ev[lane], fx_off = fr_fx[lane], hp_off = fr_hp[lane];
  if (tag == 'I' || tag == 'U' || tag == 'D') {
    const DevSlot& s = p.slots[row_slot];
    const DevCol* cols = p.cols + s.cols_base;
    u8* body = pg.fixed + fx_off;
    uint32_t flags = tag != 'I' ? old_kind : 0u;
    uint32_t st_new[SW];
    for (int w = 0; w < SW; w++) st_new[w] = fr_st[SW + w][lane];
    uint32_t toast = fr_toast[lane];
    const uint32_t e0 = fr_err[lane];
    // 'u' cells of the new row: alias the aligned old value, else MISSING (codec/event.rs:962-974)
    while (toast) {
      const uint32_t k = __builtin_ctz(toast);
      toast &= toast - 1;
      const DevCol col = cols[k];
      uint32_t* dst = (uint32_t*)(body + old_sz + col.off_full);
      const bool from_full = old_kind == ETLG_OLD_FULL, from_key = old_kind == ETLG_OLD_KEY && col.identity;
      uint32_t cst;
      if (from_full || from_key) {
        const uint32_t* src = (const uint32_t*)(body + (from_full ? col.off_full : col.off_key));
        const uint32_t nw = slot_bytes(col.cls) >> 2;
        for (uint32_t w = 0; w < nw; w++) dst[w] = src[w];
        const uint32_t oi = from_full ? k : (uint32_t)col.key_index;
        cst = (fr_st[WIDE ? oi >> 4 : 0u][lane] >> (2 * (WIDE ? oi & 15u : oi))) & 3u;
      } else {
        slot_zero(dst, col.cls);
        cst = (uint32_t)ETLG_CELL_MISSING;
        flags |= ETLG_FLAG_PARTIAL;
      }
      if constexpr (WIDE) { if (k >= 16) st_new[SW - 1] |= cst << (2 * (k & 15u)); else st_new[0] |= cst << (2 * k); }
      else st_new[0] |= cst << (2 * k);
    }
    if (e0 != 0xFFFFFFFFu) { record_error(pg, f, RK_DECODE, e0 & 0xFF); return; }
    {  // the 2-bit states at the head of each row image: st_key / st_full bytes (4 per 16 columns)
      const uint32_t so = old_kind == ETLG_OLD_NONE ? 0u : old_kind == ETLG_OLD_KEY ? s.st_key : s.st_full;
      for (uint32_t w = 0; 4 * w < so && w < (uint32_t)SW; w++) ((uint32_t*)body)[w] = fr_st[w][lane];
      if (tag != 'D') for (uint32_t w = 0; 4 * w < s.st_full && w < (uint32_t)SW; w++) ((uint32_t*)(body + old_sz))[w] = st_new[w];
    }
    pg.ev_kind[ev_idx] = (u8)tag;
    pg.ev_flags[ev_idx] = (u8)flags;
    pg.ev_table[ev_idx] = rel_id;
    pg.ev_slot[ev_idx] = s.host_id;
    pg.ev_start[ev_idx] = COPY ? 0ull : ld_be64(v.fr + 6);   // (the Insert frames of the row -> frame rewrite carry a zero WAL position)
    pg.ev_commit[ev_idx] = tx.final_lsn;
    pg.ev_ord[ev_idx] = tx.ord;
    pg.ev_body[ev_idx] = fx_off;
  } else {
    RowMsg dummy{};
    write_frame(p, v, tx, dummy, -1, ev_idx, fx_off, hp_off, nullptr, use_lds);
  }
  TSTAMP(8);
}

#ifndef ETLG_CELLS_MINBLOCKS
#define ETLG_CELLS_MINBLOCKS 4   // workgroups per CU the register allocator leaves room for: 4 x 4 waves = 128 VGPRs (24 spilled). With the
                                 // one-dword cell table a 12-column tile is ~38 KB of LDS, so four fit; measured on cfg3: 281 us at three
                                 // workgroups (168 VGPRs, no spills), 238 us at four (profiles/r02u_cells_variants.json)
#endif
// COPYK: the tiles are 64 table-copy rows each (etlg_copy_decode): `pg.in` / `pg.offs` are the COPY text rows and their offsets, the
// cell table has two dwords per cell, and the splitter of cells_tile<.., 2, ..> takes P1's place — the rows reach the arena in one
// kernel, without being rewritten as Insert frames first (copy.hip stays as the path for batches with a malformed row).
template <int NW, bool WIDE, bool COPYK>
__global__ __launch_bounds__(NW * 64, ETLG_CELLS_MINBLOCKS) void k_cells(DecParams pg, FusedParams q) {
  ETLG_DYNAMIC_LDS(smem);
  __shared__ uint32_t s_offs[CF + 1];
  __shared__ int32_t fr_slot[CF];
  __shared__ uint32_t fr_meta[CF];
  __shared__ uint32_t fr_n[CF];      // n_old | n_new << 16
  __shared__ uint64_t fr_fx[CF];     // fixed-arena offset of the frame's body
  __shared__ uint32_t fr_hp[CF];     // heap offset of the frame's first entry
  __shared__ uint32_t fr_ev[CF];     // index of the frame's event
  __shared__ uint32_t fr_st[WIDE ? 4 : 2][CF];  // 2-bit cell states of the old / new row: one word per 16 columns and image
  __shared__ uint32_t fr_err[CF];    // min over (order << 8 | code)
  __shared__ uint32_t fr_toast[CF];  // new-row columns sent as 'u'
  __shared__ uint32_t s32[16];
  __shared__ uint64_t s64[8];
  __shared__ uint8_t vlist[2][64];
  __shared__ uint64_t fcache[COPYK ? kFloatCache : 1][COPYK ? CF : 1];   // (table-copy tiles only)
  __shared__ uint64_t fc_ok[COPYK ? kFloatCache : 1];
  __shared__ uint8_t vinv[COPYK ? 64 : 1];
  __shared__ uint32_t cxm[COPYK ? CF : 1];
  const uint32_t tid = threadIdx.x;
  if (q.clear_words) {  // descriptors are double buffered: this launch clears the buffer the next batch will use
    const uint32_t per = (q.clear_words + gridDim.x - 1) / gridDim.x;
    for (uint32_t i = tid; i < per; i += NW * 64) { const uint32_t w = blockIdx.x * per + i; if (w < q.clear_words) q.d_clear[w] = 0; }
  }
  if (!(pg.flags & 16u) && !load_carry(pg)) return;  // ASYNC chain: the state the batch before this one left (flags bit 4: read late, by the tiles that need it — txn_lookback)
  if constexpr (COPYK) { if (blockIdx.x == 0 && tid == 0) pg.res->copy_span = pg.offs[pg.nframes] - pg.offs[0]; }
  DecParams p = pg;
  uint32_t dbg_u;
  ETLG_SCALAR_COPY(dbg_u, q.dbg);
  if ((dbg_u & 8) && ((tid >> 6) + ((dbg_u & 0x10000u) ? 0u : blockIdx.x)) % NW == 0 && (tid & 63) == 0) s64[7] = clock64();
  if (tid < 3) s64[tid] = 0;
  if (tid < 2) s32[8 + tid] = 0;  // column queues of P2 / P3
  // ---- P0: side tables, offsets, staging
  SideRegs side;  // variant head (lookback.hip.h): one round trip for all four tables, LDS stores after the staging loads
  side_load<NW * 64>(p, true, (uint32_t*)smem, tid, side);
  const uint32_t maxc = q.maxc, VC = 2 * maxc;
  // dynamic LDS: side tables | cell table (one dword per cell) | staging window; a tile read in place spreads its
  // three-dword cells over table + window (the host sizes the allocation for that, etlg_k_cells_lds_floor)
  uint32_t* ct = (uint32_t*)(smem + q.side_bytes);  // side_bytes is a multiple of 16
  const uint32_t table_bytes = VC * CF * (COPYK ? 8u : 4u);
  u8* stage = (u8*)ct + table_bytes;
  const uint32_t tile = blockIdx.x;
  const uint32_t lane = tid & 63;
  const int wave = (int)(((tid >> 6) + ((dbg_u & 0x10000u) ? 0u : tile)) & (uint32_t)(NW - 1));   // role of this wave in the tile (cells_tile)
  const uint32_t f0 = tile * CF;
  uint32_t nt = pg.nframes - f0 < (uint32_t)CF ? pg.nframes - f0 : (uint32_t)CF;
  // the tile's byte span from two scalar loads: staging starts while the per-frame offsets are in flight
  const ETLG_CONST_AS uint32_t* offs_c = (const ETLG_CONST_AS uint32_t*)(uintptr_t)pg.offs;  // real scalar loads (see k_fused)
  const uint32_t span0 = offs_c[f0], span1 = offs_c[f0 + nt];
  const uint32_t my_o = tid <= nt ? pg.offs[f0 + tid] : 0u;
  const uint32_t a0 = span0 & ~15u;
  // (table-copy tiles keep three bitmaps of the window behind it, 3/8 of its bytes: etlg_k_copy_cells_lds)
  const uint32_t avail = q.lds_bytes - q.side_bytes;
  const uint32_t wcap = COPYK ? (avail > table_bytes + 128 ? (((avail - table_bytes - 128) / 11) * 8) & ~15u : 0u) : (avail > table_bytes ? avail - table_bytes : 0u);
  const bool window_ok = q.in_aligned && span1 > span0 && span1 <= pg.in_len && span1 - a0 + 16 <= kWinMax && (uint64_t)(span1 - a0) + 16 <= wcap;
  if (window_ok) {
    const uint32_t full_end = a0 + ((span1 - a0) & ~15u);
    stage_chunks<NW * 64>(pg.in, stage, a0, full_end, tid);
    for (uint32_t c = full_end + tid; c < span1; c += NW * 64) stage[c - a0] = pg.in[c];
  }
  side_store<NW * 64>((uint32_t*)smem, tid, side);
  if (tid <= nt) s_offs[tid] = my_o;  // CF + 1 <= NW * 64 entries
  __syncthreads();
  TSTAMP(0);
  bool lane_ok = true;
  if (tid < nt) {
    const uint32_t o0 = s_offs[tid], o1 = s_offs[tid + 1];
    lane_ok = COPYK ? (o1 >= o0 && o0 >= span0 && o1 <= span1) : (o1 <= o0 || o1 > pg.in_len || (o0 >= span0 && o1 <= span1));
  }
  const bool use_lds = __syncthreads_and(lane_ok ? 1 : 0) && window_ok;
  TSTAMP(1);
  if constexpr (COPYK) {
    // table_row.rs:51 validates a row as UTF-8 before anything else. Every row that decodes here ends in a newline, so the rows of the
    // tile are valid exactly when the tile's bytes are, and those are checked four at a time by the whole workgroup (position-wise
    // rule, utf8_swar.h). A tile that fails — or whose rows do not fit the window — marks all its rows (cells_tile).
    uint32_t bad8 = use_lds ? 0u : 1u;
    uint32_t* const bm_sep = (uint32_t*)(stage + wcap);
    uint32_t* const bm_bs = bm_sep + (wcap / 32 + 2);
    uint32_t* const bm_nl = bm_bs + (wcap / 32 + 2);
    const uint32_t nchunks = use_lds ? (span1 - a0 + 15) / 16 + 2 : 0u;   // (+2: look-aheads read one word past the last row)
    if (use_lds) {
      // where the special bytes are: one bit per window byte, 16 bytes per thread and step
      for (uint32_t ci = tid; ci < nchunks; ci += NW * 64) {
        const uint4 x4 = *(const uint4*)(stage + 16 * ci);
        const uint32_t xs[4] = {x4.x, x4.y, x4.z, x4.w};
        uint32_t sep16 = 0, bs16 = 0, nl16 = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          auto eq = [](uint32_t v4, uint32_t c) { const uint32_t t = v4 ^ (c * 0x01010101u); return ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) & 0x80808080u; };   // bit 7 of every byte equal to c (exact)
          auto nib = [](uint32_t m) { return (((m >> 7) * 0x01020408u) >> 24) & 0xFu; };   // those four bits side by side
          const uint32_t b = eq(xs[u], '\\'), nl = eq(xs[u], '\n'), sp = nl | eq(xs[u], '\t');
          sep16 |= nib(sp) << (4 * u); bs16 |= nib(b) << (4 * u); nl16 |= nib(nl) << (4 * u);
        }
        ((uint16_t*)bm_sep)[ci] = (uint16_t)sep16;
        ((uint16_t*)bm_bs)[ci] = (uint16_t)bs16;
        ((uint16_t*)bm_nl)[ci] = (uint16_t)nl16;
      }
      __syncthreads();
      // a tab / newline behind an odd run of backslashes is an escaped character, not a separator (the run cannot reach back into
      // the row before: a row that decodes here ends in a newline)
      for (uint32_t w = tid; 2 * w < nchunks; w += NW * 64) {
        uint32_t esc = bm_sep[w] & ((bm_bs[w] << 1) | (w ? bm_bs[w - 1] >> 31 : 0u));
        uint32_t clear = 0;
        while (esc) {
          const uint32_t bit = (uint32_t)__builtin_ctz(esc);
          esc &= esc - 1u;
          uint32_t qb = (w << 5) + bit, run = 0;   // count the backslashes that end at qb - 1
          while (qb > 0 && ((bm_bs[(qb - 1) >> 5] >> ((qb - 1) & 31u)) & 1u)) { run++; qb--; }
          if (run & 1u) clear |= 1u << bit;
        }
        if (clear) bm_sep[w] &= ~clear;
      }
      const u8* wp = stage + (span0 - a0);
      const uint32_t nb = span1 - span0;
      for (uint32_t d = tid; 4 * d < nb; d += NW * 64) {
        const uint32_t rem = nb - 4 * d;
        uint32_t cur = ldu32(wp + 4 * d);
        if (rem < 4) cur &= (1u << (8 * rem)) - 1u;
        const uint32_t prev = d ? ldu32(wp + 4 * d - 4) : 0u;
        if (((cur | prev) & 0x80808080u) && utf8_dword_bad(prev, cur, rem == 4)) bad8 = 1;
      }
    }
    const uint32_t copy_bad = __syncthreads_and(bad8 ? 0 : 1) ? 0u : 1u;
    const CellsLds shc{s_offs, fr_slot, fr_meta, fr_n, fr_fx, fr_hp, fr_ev, fr_st, fr_err, fr_toast, s32, s64, ct, vlist, bm_sep, bm_bs, bm_nl, cxm, fcache, fc_ok, vinv};
    cells_tile<NW, 2, WIDE>(p, pg, q, shc, stage, a0, tile, nt, copy_bad);
  }
  if constexpr (!COPYK) {
    const CellsLds sh{s_offs, fr_slot, fr_meta, fr_n, fr_fx, fr_hp, fr_ev, fr_st, fr_err, fr_toast, s32, s64, ct, vlist, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // frames are addressed as base + (offset - b0): the LDS window, or (tiles that do not fit) the input itself
    if (use_lds) cells_tile<NW, 1, WIDE>(p, pg, q, sh, stage, a0, tile, nt);
    else cells_tile<NW, 0, WIDE>(p, pg, q, sh, pg.in, 0u, tile, nt);
  }
}

}  // namespace etlg

extern "C" {

using namespace etlg;

void etlg_k_launch_cells(const DecParams* p, const void* qv, hipStream_t s) {
  const FusedParams* q = (const FusedParams*)qv;
  if (q->maxc > (uint32_t)MAXC_NARROW) hipLaunchKernelGGL((k_cells<4, true, false>), dim3(q->ntiles), dim3(256), q->lds_bytes, s, *p, *q);
  else hipLaunchKernelGGL((k_cells<4, false, false>), dim3(q->ntiles), dim3(256), q->lds_bytes, s, *p, *q);
}

// table-copy rows straight into the arena (p->in / p->offs: the rows and their offsets; q->copy_rel: the table's id)
void etlg_k_launch_copy_cells(const DecParams* p, const void* qv, hipStream_t s) {
  const FusedParams* q = (const FusedParams*)qv;
  if (q->maxc > (uint32_t)MAXC_NARROW) hipLaunchKernelGGL((k_cells<4, true, true>), dim3(q->ntiles), dim3(256), q->lds_bytes, s, *p, *q);
  else hipLaunchKernelGGL((k_cells<4, false, true>), dim3(q->ntiles), dim3(256), q->lds_bytes, s, *p, *q);
}

int etlg_k_cells_set_lds(void) {
  const hipError_t a = hipFuncSetAttribute((const void*)k_cells<4, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 6144);
  const hipError_t b = hipFuncSetAttribute((const void*)k_cells<4, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 6656);
  const hipError_t a2 = hipFuncSetAttribute((const void*)k_cells<4, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 6144);
  const hipError_t b2 = hipFuncSetAttribute((const void*)k_cells<4, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 6656);
  return a == hipSuccess && b == hipSuccess && a2 == hipSuccess && b2 == hipSuccess ? 0 : 1;
}

uint32_t etlg_k_cells_table_bytes(uint32_t maxc) { return 2u * maxc * CF * 4u; }       // next to the window of a staged tile
uint32_t etlg_k_copy_cells_table_bytes(uint32_t maxc) { return 2u * maxc * CF * 8u; }  // ... of a tile of table-copy rows
// dynamic LDS of a table-copy tile whose window holds `window` bytes: cell table, window, three bitmaps of the window (k_cells<.., COPYK>)
uint32_t etlg_k_copy_cells_lds(uint32_t maxc, uint32_t window) { return 2u * maxc * CF * 8u + ((window + 15u) & ~15u) / 8u * 11u + 192u; }
uint32_t etlg_k_cells_lds_floor(uint32_t maxc) { return 3u * 2u * maxc * CF * 4u; }    // table + window together: what a tile read in place needs
uint32_t etlg_k_cells_static_lds(uint32_t maxc) { return maxc > (uint32_t)MAXC_NARROW ? 4224u : 3648u; }  // the kernel's __shared__ arrays (WAL tiles: 3 6xx / 4 1xx bytes in the gfx950 build of the narrow / wide instantiation) + slack
uint32_t etlg_k_copy_cells_static_lds(uint32_t maxc) { return (maxc > (uint32_t)MAXC_NARROW ? 4224u : 3648u) + kFloatCache * (CF + 1) * 8u + 64u + 256u + 64u; }  // ... of a table-copy tile (+ float cache, column masks)
uint32_t etlg_k_cells_maxc(void) { return MAXC_WIDE; }

}  // extern "C"
