// Table-copy rows for gfx950 (include/etlg.h, etlg_copy_decode): COPY ... TO STDOUT text rows ->
// the same arena the Insert path produces, through the same value codec.
//
// Since round 3 a batch of rows first goes through k_copy_cells (cells.hip: rows -> arena in one kernel).
// This file is the path behind it — batches with a malformed row (the reference's error order is
// decided here), rows that do not fit a tile's LDS window, tables wider than k_cells takes,
// ETLG_COPY_DIRECT=0 — and the reference point the one-kernel path is tested against.
//
// parse_table_row_from_postgres_copy_bytes (crates/etl/src/postgres/codec/table_row.rs:47-254) is
// a field splitter + unescaper in front of parse_cell_from_postgres_text. This kernel is that
// front end: one lane per row (256 rows per workgroup, the rows staged into LDS with coalesced
// loads) validates the row as UTF-8, splits it at unescaped tabs / newlines, undoes the backslash
// escapes and rewrites the row as a pgoutput Insert frame
//     'd' len 'w' 0{24} 'I' rel_id 'N' ncols { 'n' | 't' len bytes }*
// in a second buffer; the decode kernels (fused.hip / cells.hip / kernels.hip) then run over
// those frames in "copy mode" (DecParams.flags bit 1: the caller's schema slot, NULL allowed in
// every column). Row i gets exactly rowlen_i + C bytes (C = 38 + 5 * ncols covers the frame
// header and the per-cell headers; unescaping only shrinks), so the frame offsets are a closed
// form of the row offsets and no scan is needed; the slack after the last cell is covered by the
// CopyData length and ignored by the tuple walk, like trailing bytes in the reference's parser.
//
// Row-level errors keep the reference's order: invalid UTF-8 comes first (rank WIRE, the row
// becomes an all-NULL tuple); "not properly terminated" / more / fewer columns come after the
// errors of the cells that were completed before them (rank COPY_SHAPE > DECODE; missing cells
// are padded with NULLs so that the completed ones still decode and report theirs).
#include "codec.hip.h"

namespace etlg {

struct CopyParams {
  const u8* rows;            // concatenated row payloads
  const uint32_t* row_offs;  // nrows + 1
  uint32_t nrows;
  uint32_t ncols;            // replicated columns of the slot
  uint64_t rows_len;
  u8* out;                   // synthetic frames
  uint32_t* out_offs;        // nrows + 1: out_offs[i] = (row_offs[i] - row_offs[0]) + i * C
  uint32_t rel_id;
  uint32_t C;                // bytes added per row
  uint32_t lds_bytes;        // dynamic LDS (staging window)
  uint32_t in_aligned;
  DecParams dec;             // for record_error (res)
};

constexpr uint32_t kCopyHdr = 38;  // 'd' len(4) 'w' 24 x 0 'I' rel(4) 'N' ncols(2)

template <bool STAGED>
DEV void copy_row(const CopyParams& q, uint32_t r, const u8* row, uint32_t n, u8* fr, uint32_t slot_bytes) {
  // frame header: 38 bytes as four 8-byte stores and three 2-byte ones (a frame starts at any byte; one byte at a time it was 38 store
  // instructions per row)
  {
    const uint32_t L = slot_bytes - 1;
    const uint64_t w0 = (uint64_t)'d' | ((uint64_t)__builtin_bswap32(L) << 8) | ((uint64_t)'w' << 40);   // bytes 0..7: 'd' len 'w' 0 0
    const uint64_t zero = 0;
    const uint64_t w3 = ((uint64_t)'I' << 48) | ((uint64_t)(q.rel_id >> 24) << 56);                      // bytes 24..31: 0 x 6, 'I', rel[0]
    const uint16_t h0 = (uint16_t)(((q.rel_id >> 16) & 0xFFu) | (((q.rel_id >> 8) & 0xFFu) << 8));         // bytes 32..33
    const uint16_t h1 = (uint16_t)((q.rel_id & 0xFFu) | ((uint32_t)'N' << 8));                            // bytes 34..35
    const uint16_t h2 = (uint16_t)(((q.ncols >> 8) & 0xFFu) | ((q.ncols & 0xFFu) << 8));                  // bytes 36..37
    __builtin_memcpy(fr, &w0, 8); __builtin_memcpy(fr + 8, &zero, 8); __builtin_memcpy(fr + 16, &zero, 8); __builtin_memcpy(fr + 24, &w3, 8);
    __builtin_memcpy(fr + 32, &h0, 2); __builtin_memcpy(fr + 34, &h1, 2); __builtin_memcpy(fr + 36, &h2, 2);
  }
  u8* o = fr + kCopyHdr;
  uint32_t col = 0;
  uint32_t err = 0;
  if (!utf8_valid(row, n)) {  // table_row.rs:51 (simdutf8 over the whole row)
    record_error(q.dec, r, RK_WIRE, ETLG_E_UTF8);
  } else {
    uint32_t pos = 0;
    bool terminated = false;
    for (;;) {
      const uint32_t field_start = pos;
      const bool keep = col < q.ncols;  // a field beyond the schema is only scanned (it is the error below)
      u8* const cell = o;
      u8* w = o + 5;
      bool ended = false;
      while (pos < n) {
        if (pos + 8 <= n) {  // eight bytes at a time while there is no tab, newline or backslash among them
          uint64_t x; __builtin_memcpy(&x, row + pos, 8);
          auto has = [](uint64_t v, uint64_t c) { const uint64_t t = v ^ (c * 0x0101010101010101ull); return (t - 0x0101010101010101ull) & ~t & 0x8080808080808080ull; };
          const uint64_t m = has(x, '\t') | has(x, '\n') | has(x, '\\');
          if (!m) {
            if (keep) { __builtin_memcpy(w, &x, 8); w += 8; }
            pos += 8;
            continue;
          }
          const uint32_t k = (uint32_t)__builtin_ctzll(m) >> 3;  // clean bytes before the first special one (exact for the lowest match)
          if (keep) for (uint32_t j = 0; j < k; j++) w[j] = (u8)(x >> (8 * j));
          if (keep) w += k;
          pos += k;
        }
        const uint32_t c = row[pos];
        if (c == '\t') { pos++; ended = true; break; }
        if (c == '\n') { pos++; ended = true; terminated = true; break; }
        if (c == '\\') {  // :129-176
          pos++;
          if (pos < n) {
            const uint32_t e = row[pos];
            if (e < 0x80u) {
              uint32_t ch = e;
              if (e == 'b') ch = 8; else if (e == 'f') ch = 12; else if (e == 'n') ch = '\n';
              else if (e == 'r') ch = '\r'; else if (e == 't') ch = '\t'; else if (e == 'v') ch = 11;
              if (keep) *w++ = (u8)ch;
              pos++;
            } else {  // a whole multi-byte character (the row is valid UTF-8)
              const uint32_t l = e >= 0xF0u ? 4u : e >= 0xE0u ? 3u : 2u;
              for (uint32_t k = 0; k < l; k++) { if (keep) *w++ = row[pos + k]; }
              pos += l;
            }
          }
          continue;
        }
        if (keep) *w++ = (u8)c;
        pos++;
      }
      if (!ended) {  // no terminator left: the dangling field is dropped (:93-106)
        if (!terminated) err = ETLG_E_COPY_UNTERMINATED;
        break;
      }
      if (!keep) { err = ETLG_E_COPY_MORE_COLS; break; }  // :179-192
      const uint32_t raw_len = pos - 1 - field_start;
      if (raw_len == 2 && row[field_start] == '\\' && row[field_start + 1] == 'N') {  // the NULL marker, matched before unescaping (:199)
        cell[0] = 'n';
        o = cell + 1;
      } else {
        const uint32_t ulen = (uint32_t)(w - (cell + 5));
        cell[0] = 't';
        { const uint32_t be = __builtin_bswap32(ulen); __builtin_memcpy(cell + 1, &be, 4); }
        o = w;
      }
      col++;
    }
    if (!err && col < q.ncols) err = ETLG_E_COPY_FEWER_COLS;  // :234-249
    if (err) record_error(q.dec, r, RK_COPY_SHAPE, err);
  }
  for (; col < q.ncols; col++) *o++ = 'n';  // keep the tuple as wide as the schema: the completed cells still decode
}

__global__ __launch_bounds__(256) void k_copy_frames(CopyParams q) {
  ETLG_DYNAMIC_LDS(smem);
  __shared__ uint32_t s_offs[257];
  const uint32_t tid = threadIdx.x;
  const uint32_t r0 = blockIdx.x * 256u;
  const uint32_t nt = q.nrows - r0 < 256u ? q.nrows - r0 : 256u;
  for (uint32_t i = tid; i <= nt; i += 256) s_offs[i] = q.row_offs[r0 + i];
  __syncthreads();
  const uint32_t base0 = q.row_offs[0];
  const uint32_t span0 = s_offs[0], span1 = s_offs[nt];
  bool lane_ok = true;
  if (tid < nt) lane_ok = s_offs[tid] <= s_offs[tid + 1] && s_offs[tid] >= span0 && s_offs[tid + 1] <= span1;
  const uint32_t a0 = span0 & ~15u;
  const bool window_ok = q.in_aligned && span1 >= span0 && span1 <= q.rows_len && (uint64_t)(span1 - a0) + 16 <= q.lds_bytes;
  const bool use_lds = __syncthreads_and(lane_ok ? 1 : 0) && window_ok;
  if (use_lds) {
    const uint32_t full_end = a0 + ((span1 - a0) & ~15u);
    for (uint32_t c = a0 + 16 * tid; c < full_end; c += 64 * 256) {
      const uint32_t c1 = c + 16 * 256, c2 = c + 32 * 256, c3 = c + 48 * 256;
      uint4 v0 = *(const uint4*)(q.rows + c), v1 = make_uint4(0, 0, 0, 0), v2 = v1, v3 = v1;
      if (c1 < full_end) v1 = *(const uint4*)(q.rows + c1);
      if (c2 < full_end) v2 = *(const uint4*)(q.rows + c2);
      if (c3 < full_end) v3 = *(const uint4*)(q.rows + c3);
      *(uint4*)(smem + (c - a0)) = v0;
      if (c1 < full_end) *(uint4*)(smem + (c1 - a0)) = v1;
      if (c2 < full_end) *(uint4*)(smem + (c2 - a0)) = v2;
      if (c3 < full_end) *(uint4*)(smem + (c3 - a0)) = v3;
    }
    for (uint32_t c = full_end + tid; c < span1; c += 256) smem[c - a0] = q.rows[c];
  }
  __syncthreads();
  if (tid >= nt) return;
  const uint32_t r = r0 + tid;
  uint32_t o0 = s_offs[tid];
  const uint32_t o1 = s_offs[tid + 1];
  const uint32_t n = o0 >= base0 && o1 >= o0 && o1 <= q.rows_len ? o1 - o0 : 0u;  // malformed offsets: an empty (unterminated) row
  if (o0 < base0 || o0 > q.rows_len) o0 = base0;                                    // ... that stays inside the output buffer
  const uint64_t f0 = (uint64_t)(o0 - base0) + (uint64_t)r * q.C;
  const uint32_t slot_bytes = n + q.C;
  q.out_offs[r] = (uint32_t)f0;
  if (r + 1 == q.nrows) q.out_offs[q.nrows] = (uint32_t)(f0 + slot_bytes);
  if (use_lds) copy_row<true>(q, r, smem + (o0 - a0), n, q.out + f0, slot_bytes);
  else copy_row<false>(q, r, q.rows + o0, n, q.out + f0, slot_bytes);
}


// ---------------------------------------------------------------- lane-per-byte splitter (round 3; selected with ETLG_COPY_KERNEL=1, not the default)
// The same rows -> frames rewrite, data-parallel: one wave takes kRpw consecutive rows and streams their bytes through the lanes,
// 64 bytes per step (lane = byte), so the cost no longer follows the longest field of a wave's rows. Everything per byte is a
// function of wave-wide ballot masks (simdjson's way):
//   * which backslashes escape: a backslash escapes iff it stands at an even distance from the start of its run of backslashes
//     (run starts: a backslash whose predecessor is not one; a row start; the byte behind an escaper carried from the previous step);
//   * separators = tabs / newlines that are not escaped; a field starts behind a separator or at a row start;
//   * the NULL marker is a field that is exactly `\N` (matched on the raw bytes, table_row.rs:199): escaper at a field start, 'N',
//     separator;
//   * output position of a content byte = frame base + 38 + 5 x (complete fields before it in its row + 1) - 4 x (NULL fields before)
//     + content bytes before it in its row; a field's cell header (`t` + be32 length, or `n`) is written by the lane that holds its
//     separator. "Before it in its row" = popcounts of the masks between the row's start and the lane, plus what the row carried in
//     from earlier steps (five wave-uniform counters).
// Row-level errors keep the reference's order (UTF-8 first — position-wise rule of utf8_swar.h, as masks —, then cells, then
// not-terminated / more / fewer columns). Rows with broken offsets take the lane-per-row path (copy_row) for their whole group.
constexpr uint32_t kRpw = 32;          // rows per wave
constexpr uint32_t kBlk = 4096;        // bytes staged per block (+ kLook bytes of look-ahead)
constexpr uint32_t kLook = 16;

DEV uint32_t clz64(uint64_t v) { return (uint32_t)__builtin_clzll(v); }
DEV uint32_t pop64(uint64_t v) { return (uint32_t)__builtin_popcountll(v); }
DEV uint32_t rl63(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }

__global__ __launch_bounds__(64) void k_copy_split(CopyParams q) {
  __shared__ __attribute__((aligned(16))) u8 win[kBlk + 64];
  __shared__ uint32_t rsbits[(kBlk + 64) / 32 + 2];
  __shared__ uint32_t roffs[kRpw + 2];
  __shared__ uint32_t nzr[kRpw + 1];      // the non-empty rows of the group, in order
  __shared__ uint32_t row_bad[kRpw + 1];  // bit 0: invalid UTF-8
  const uint32_t lane = threadIdx.x;
  const uint32_t r0 = blockIdx.x * kRpw;
  const uint32_t nr = q.nrows - r0 < kRpw ? q.nrows - r0 : kRpw;
  const uint32_t base0 = q.row_offs[0];
  if (lane <= nr) roffs[lane] = q.row_offs[r0 + lane];
  if (lane < kRpw + 1) row_bad[lane] = 0;
  __syncthreads();   // (a workgroup is one wave; the emulator's lanes are fibers and need the rendezvous)
  const uint32_t o0 = lane < nr ? roffs[lane] : 0u, o1 = lane < nr ? roffs[lane + 1] : 0u;
  const bool off_ok = lane >= nr || (o0 <= o1 && o0 >= base0 && o1 <= q.rows_len);
  if (__ballot(!off_ok)) {  // malformed offsets somewhere in the group: one lane per row, rows read in place (copy_row's rules)
    if (lane < nr) {
      const uint32_t r = r0 + lane;
      uint32_t a = o0;
      const uint32_t n = a >= base0 && o1 >= a && o1 <= q.rows_len ? o1 - a : 0u;
      if (a < base0 || a > q.rows_len) a = base0;
      const uint64_t f0 = (uint64_t)(a - base0) + (uint64_t)r * q.C;
      q.out_offs[r] = (uint32_t)f0;
      if (r + 1 == q.nrows) q.out_offs[q.nrows] = (uint32_t)(f0 + n + q.C);
      copy_row<false>(q, r, q.rows + a, n, q.out + f0, n + q.C);
    }
    return;
  }
  // ---- per row (lane = row): frame header, frame offset, the list of non-empty rows; empty rows are finished here
  const uint32_t n_mine = o1 - o0;
  const unsigned long long nzm = __ballot(lane < nr && n_mine != 0);
  if (lane < nr && n_mine) nzr[pop64(nzm & ((1ull << lane) - 1))] = lane;
  const uint32_t n_nz = pop64(nzm);
  __syncthreads();
  if (lane < nr) {
    const uint32_t r = r0 + lane;
    const uint64_t f0 = (uint64_t)(o0 - base0) + (uint64_t)r * q.C;
    q.out_offs[r] = (uint32_t)f0;
    if (r + 1 == q.nrows) q.out_offs[q.nrows] = (uint32_t)(f0 + n_mine + q.C);
    u8* fr = q.out + f0;
    const uint32_t L = n_mine + q.C - 1;
    const uint64_t w0 = (uint64_t)'d' | ((uint64_t)__builtin_bswap32(L) << 8) | ((uint64_t)'w' << 40);
    const uint64_t zero = 0;
    const uint64_t w3 = ((uint64_t)'I' << 48) | ((uint64_t)(q.rel_id >> 24) << 56);
    const uint16_t h0 = (uint16_t)(((q.rel_id >> 16) & 0xFFu) | (((q.rel_id >> 8) & 0xFFu) << 8));
    const uint16_t h1 = (uint16_t)((q.rel_id & 0xFFu) | ((uint32_t)'N' << 8));
    const uint16_t h2 = (uint16_t)(((q.ncols >> 8) & 0xFFu) | ((q.ncols & 0xFFu) << 8));
    __builtin_memcpy(fr, &w0, 8); __builtin_memcpy(fr + 8, &zero, 8); __builtin_memcpy(fr + 16, &zero, 8); __builtin_memcpy(fr + 24, &w3, 8);
    __builtin_memcpy(fr + 32, &h0, 2); __builtin_memcpy(fr + 34, &h1, 2); __builtin_memcpy(fr + 36, &h2, 2);
    if (!n_mine) {  // an empty row: no field, no terminator
      record_error(q.dec, r, RK_COPY_SHAPE, ETLG_E_COPY_UNTERMINATED);
      for (uint32_t k = 0; k < q.ncols; k++) fr[kCopyHdr + k] = 'n';
    }
  }
  if (!n_nz) return;
  const uint32_t g0 = roffs[nzr[0]], g1 = roffs[nr];
  // ---- the stream: carried state of the row / field / escape that is open at the end of a step (wave-uniform)
  uint32_t c_esc = 0, c_sep = 0, c_nn = 0, c_ns = 0, c_must = 0;          // bits carried into the next step's low lanes
  uint32_t c_fields = 0, c_cont = 0, c_nulls = 0, c_term = 0, c_open = 0; // counters of the open row / field
  uint32_t nz_before = 0;                                                  // non-empty rows started before the step
  const uint64_t lt = (1ull << lane) - 1, le = lt | (1ull << lane);
  for (uint32_t blk0 = g0; blk0 < g1; blk0 += kBlk) {
    // stage [blk0, blk0 + kBlk + kLook) (what exists of it) and mark the row starts that fall into the block
    for (uint32_t i = lane * 16; i < kBlk + kLook; i += 64 * 16) {
      const uint64_t a = (uint64_t)blk0 + i;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (a + 16 <= q.rows_len) __builtin_memcpy(&v, q.rows + a, 16);
      else { u8 t[16]; for (uint32_t b = 0; b < 16; b++) t[b] = a + b < q.rows_len ? q.rows[a + b] : (u8)0; __builtin_memcpy(&v, t, 16); }
      *(uint4*)(win + i) = v;
    }
    for (uint32_t i = lane; i < (kBlk + 64) / 32 + 2; i += 64) rsbits[i] = 0;
    __syncthreads();   // (a workgroup is one wave here)
    if (lane < nr && n_mine && o0 >= blk0 && o0 - blk0 < kBlk) atomicOr(&rsbits[(o0 - blk0) >> 5], 1u << ((o0 - blk0) & 31u));
    __syncthreads();
    const uint32_t bend = g1 - blk0 < kBlk ? g1 - blk0 : kBlk;
    for (uint32_t c0 = 0; c0 < bend; c0 += 64) {
      const uint32_t w = c0 + lane;                 // position inside the block
      const uint32_t p = blk0 + w;                  // ... inside the rows buffer
      const bool valid = w < bend;
      const uint32_t c = valid ? win[w] : 0u, c1 = win[w + 1], c2 = win[w + 2];
      const uint64_t V = __ballot(valid);
      const uint64_t R = __ballot(valid && ((rsbits[w >> 5] >> (w & 31u)) & 1u));
      // ---- escapes
      const uint64_t BS = __ballot(c == '\\');
      const uint64_t cin = (uint64_t)c_esc & ~R & 1ull;
      const uint64_t bsp = BS & ~cin;
      const uint64_t starts = (bsp & ~(bsp << 1)) | (bsp & R);
      const uint64_t ms = starts & le;
      const bool escaper = ((bsp >> lane) & 1ull) && ms && (((lane - (63u - clz64(ms))) & 1u) == 0);
      const uint64_t ESC = __ballot(escaper);
      const uint64_t ESCD = ((ESC << 1) | cin) & ~R & V;
      // ---- separators, field starts, NULL markers
      const uint64_t TN = __ballot(c == '\t' || c == '\n'), NL = __ballot(c == '\n');
      const uint64_t SEP = TN & ~ESCD & V, NLS = NL & ~ESCD & V;
      const uint64_t F = ((SEP << 1) | (uint64_t)(c_sep & 1u)) | R;
      // this lane's row
      const uint64_t rm = R & le;
      const uint32_t idx = nz_before + pop64(rm) - 1u;          // (the stream starts at a row start: never negative for a valid lane)
      const uint32_t row = valid ? nzr[idx < kRpw ? idx : 0u] : 0u;
      const uint32_t row_o0 = roffs[row], row_end = roffs[row + 1];
      const bool nq = ((F >> lane) & 1ull) && escaper && c1 == 'N' && (c2 == '\t' || c2 == '\n') && p + 2 < row_end;
      const uint64_t NQ = __ballot(nq);
      const uint64_t NN = (NQ << 1) | (uint64_t)(c_nn & 1u);
      const uint64_t NSEP = ((NQ << 2) | (uint64_t)(c_ns & 3u)) & SEP;
      const uint64_t CONT = V & ~ESC & ~SEP & ~NN;
      // ---- counts before this lane in its row
      const bool row_here = rm != 0;
      const uint32_t rs = row_here ? 63u - clz64(rm) : 0u;
      const uint64_t inrow = lt & ~((1ull << rs) - 1ull);
      const uint32_t fld = pop64(SEP & inrow) + (row_here ? 0u : c_fields);
      const uint64_t KEEP = __ballot(fld < q.ncols);
      const uint64_t CONTK = CONT & KEEP, SEPK = SEP & KEEP, NSEPK = NSEP & KEEP;
      const uint32_t cnt = pop64(CONTK & inrow) + (row_here ? 0u : c_cont);
      const uint32_t nul = pop64(NSEPK & inrow) + (row_here ? 0u : c_nulls);
      const uint32_t term = (pop64(NLS & inrow) ? 1u : 0u) | (row_here ? 0u : c_term);
      const uint64_t fm = F & le;
      const bool fld_here = fm != 0;
      const uint32_t fs = fld_here ? 63u - clz64(fm) : 0u;
      const uint32_t open = pop64(CONT & lt & ~((1ull << fs) - 1ull)) + (fld_here ? 0u : c_open);   // content of the lane's field before the lane
      const bool is_sep = (SEP >> lane) & 1ull, is_cont = (CONTK >> lane) & 1ull, is_nsep = (NSEPK >> lane) & 1ull;
      u8* const fr = q.out + ((uint64_t)(row_o0 - base0) + (uint64_t)(r0 + row) * q.C);
      u8* const cells = fr + kCopyHdr + cnt - 4u * nul;          // where the cells stand that are complete before this lane, plus the content so far
      if (is_cont) {
        uint32_t ch = c;
        if ((ESCD >> lane) & 1ull) {  // :129-176
          if (c == 'b') ch = 8; else if (c == 'f') ch = 12; else if (c == 'n') ch = '\n';
          else if (c == 'r') ch = '\r'; else if (c == 't') ch = '\t'; else if (c == 'v') ch = 11;
        }
        cells[5u * (fld + 1u)] = (u8)ch;
      }
      if (is_sep && fld < q.ncols) {  // the header of the field this separator ends
        u8* h = cells - open + 5u * fld;
        if (is_nsep) h[0] = 'n';
        else { h[0] = 't'; h[1] = (u8)(open >> 24); h[2] = (u8)(open >> 16); h[3] = (u8)(open >> 8); h[4] = (u8)open; }
      }
      // ---- UTF-8, position-wise (utf8_swar.h): a byte is a continuation byte exactly when a lead byte 1..3 places before reaches it
      {
        const uint64_t L2 = __ballot(c >= 0xC2u && c <= 0xDFu), L3 = __ballot(c >= 0xE0u && c <= 0xEFu), L4 = __ballot(c >= 0xF0u && c <= 0xF4u);
        const uint64_t CB = __ballot((c & 0xC0u) == 0x80u);
        const uint64_t must_raw = ((L2 | L3 | L4) << 1) | ((L3 | L4) << 2) | (L4 << 3) | (uint64_t)(c_must & 7u);
        const bool bad2 = (c == 0xE0u && c1 < 0xA0u) || (c == 0xEDu && c1 > 0x9Fu) || (c == 0xF0u && c1 < 0x90u) || (c == 0xF4u && c1 > 0x8Fu);
        const bool bad_here = valid && ((((must_raw & ~R) ^ CB) >> lane) & 1ull || c == 0xC0u || c == 0xC1u || c >= 0xF5u || bad2);
        if (bad_here) atomicOr(&row_bad[idx < kRpw ? idx : 0u], 1u);
        if (valid && ((must_raw & R) >> lane) & 1ull && idx > 0) atomicOr(&row_bad[idx - 1u], 1u);   // a sequence cut off by the end of the row before
        c_must = (uint32_t)((((L2 | L3 | L4) >> 63) | ((L3 | L4) >> 62) | (L4 >> 61)) & 7ull);
        if (c0 + 64 >= bend && blk0 + kBlk >= g1 && c_must && lane == 0) atomicOr(&row_bad[n_nz - 1u], 1u);          // ... by the end of the last row
      }
      // ---- the end of a row: shape errors, NULL padding for the cells that are missing
      if (valid && p + 1 == row_end) {
        const uint32_t nf = fld + (is_sep ? 1u : 0u);
        const bool terminated = term || ((NLS >> lane) & 1ull);
        const uint32_t err = nf > q.ncols ? (uint32_t)ETLG_E_COPY_MORE_COLS : !terminated ? (uint32_t)ETLG_E_COPY_UNTERMINATED
                           : nf < q.ncols ? (uint32_t)ETLG_E_COPY_FEWER_COLS : 0u;
        if (err) record_error(q.dec, r0 + row, RK_COPY_SHAPE, err);
        if (nf < q.ncols) {  // behind the last complete cell (a dangling field is dropped)
          const uint32_t dangling = is_sep ? 0u : open + (((CONT >> lane) & 1ull) ? 1u : 0u);
          const uint32_t cnt_all = cnt + (is_cont ? 1u : 0u), nul_all = nul + (is_nsep ? 1u : 0u);
          u8* pad = fr + kCopyHdr + (cnt_all - dangling) - 4u * nul_all + 5u * nf;
          for (uint32_t k = nf; k < q.ncols; k++) *pad++ = 'n';
        }
      }
      // ---- what the open row / field / escape hands to the next step
      {
        const uint32_t s63 = (uint32_t)(SEP >> 63) & 1u;
        c_fields = rl63(fld) + s63;
        c_cont = rl63(cnt) + ((uint32_t)(CONTK >> 63) & 1u);
        c_nulls = rl63(nul) + ((uint32_t)(NSEPK >> 63) & 1u);
        c_term = rl63(term) | ((uint32_t)(NLS >> 63) & 1u);
        c_open = s63 ? 0u : rl63(open) + ((uint32_t)(CONT >> 63) & 1u);
        c_esc = (uint32_t)(ESC >> 63) & 1u;
        c_sep = s63;
        c_nn = (uint32_t)(NQ >> 63) & 1u;
        c_ns = (uint32_t)(NQ >> 62) & 3u;
        nz_before += pop64(R);
      }
    }
    __syncthreads();
  }
  // ---- rows that are not valid UTF-8: the error of the row (rank WIRE: before its cells'), an all-NULL tuple
  __syncthreads();
  if (lane < n_nz && (row_bad[lane] & 1u)) {
    const uint32_t row = nzr[lane];
    record_error(q.dec, r0 + row, RK_WIRE, ETLG_E_UTF8);
    u8* fr = q.out + ((uint64_t)(roffs[row] - base0) + (uint64_t)(r0 + row) * q.C);
    for (uint32_t k = 0; k < q.ncols; k++) fr[kCopyHdr + k] = 'n';
  }
}

}  // namespace etlg

extern "C" {

using namespace etlg;

uint32_t etlg_k_copy_bytes_per_row(uint32_t ncols) { return kCopyHdr + 5u * ncols; }

int etlg_k_copy_set_lds(void) {
  return hipFuncSetAttribute((const void*)k_copy_frames, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096) == hipSuccess ? 0 : 1;
}

// rows / row_offs / out / out_offs are device pointers; `dec` only needs its `res` member.
void etlg_k_launch_copy(const uint8_t* rows, const uint32_t* row_offs, uint32_t nrows, uint64_t rows_len, uint32_t ncols,
                        uint32_t rel_id, uint8_t* out, uint32_t* out_offs, uint32_t lds_bytes, int lane_per_byte, const DecParams* dec, hipStream_t s) {
  CopyParams q;
  q.rows = rows; q.row_offs = row_offs; q.nrows = nrows; q.ncols = ncols; q.rows_len = rows_len;
  q.out = out; q.out_offs = out_offs; q.rel_id = rel_id; q.C = etlg_k_copy_bytes_per_row(ncols);
  q.lds_bytes = lds_bytes; q.in_aligned = ((uintptr_t)rows & 15) == 0;
  q.dec = *dec;
  // lane_per_byte (ETLG_COPY_KERNEL=1, read per context): the data-parallel splitter. Measured (profiles/r03w_copy_development.txt): it wins on
  // escape-heavy rows (544 vs 602 us per 78 MB) and loses on ordinary text (444 vs 184 us), where the lane-per-row kernel moves eight
  // bytes per step — so the lane-per-row kernel stays the default.
  if (lane_per_byte) hipLaunchKernelGGL(k_copy_split, dim3((nrows + kRpw - 1) / kRpw), dim3(64), 0, s, q);
  else hipLaunchKernelGGL(k_copy_frames, dim3((nrows + 255) / 256), dim3(256), lds_bytes, s, q);
}

}  // extern "C"
