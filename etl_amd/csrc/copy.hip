// Table-copy rows for gfx950 (include/etlg.h, etlg_copy_decode): COPY ... TO STDOUT text rows ->
// the same arena the Insert path produces, through the same value codec.
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

}  // namespace etlg

extern "C" {

using namespace etlg;

uint32_t etlg_k_copy_bytes_per_row(uint32_t ncols) { return kCopyHdr + 5u * ncols; }

int etlg_k_copy_set_lds(void) {
  return hipFuncSetAttribute((const void*)k_copy_frames, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096) == hipSuccess ? 0 : 1;
}

// rows / row_offs / out / out_offs are device pointers; `dec` only needs its `res` member.
void etlg_k_launch_copy(const uint8_t* rows, const uint32_t* row_offs, uint32_t nrows, uint64_t rows_len, uint32_t ncols,
                        uint32_t rel_id, uint8_t* out, uint32_t* out_offs, uint32_t lds_bytes, const DecParams* dec, hipStream_t s) {
  CopyParams q;
  q.rows = rows; q.row_offs = row_offs; q.nrows = nrows; q.ncols = ncols; q.rows_len = rows_len;
  q.out = out; q.out_offs = out_offs; q.rel_id = rel_id; q.C = etlg_k_copy_bytes_per_row(ncols);
  q.lds_bytes = lds_bytes; q.in_aligned = ((uintptr_t)rows & 15) == 0;
  q.dec = *dec;
  hipLaunchKernelGGL(k_copy_frames, dim3((nrows + 255) / 256), dim3(256), lds_bytes, s, q);
}

}  // extern "C"
