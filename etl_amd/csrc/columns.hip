// Columnar hand-off on the device: the rows of ONE schema slot out of a decoded batch, as Arrow-layout
// column buffers — validity bitmap + values (+ 64-bit offsets for var-len columns) — built by kernels over the arena that
// is already in HBM. Replaces the per-Cell builders of the reference's sinks (rows_to_record_batch / build_array_for_field,
// crates/etl-destinations/src/iceberg/encoding.rs:34-84; cell_to_* converters :150-360) with one strided gather per column:
//
//   Bool -> Boolean (bit-packed) | I16, I32 -> Int32 | I64, U32 -> Int64 (cell_to_i32 / cell_to_i64, :157-171)
//   F32 -> Float32 | F64 -> Float64 | Date -> Date32 (days since 1970-01-01, :194) | Time -> Time64(us) (:201)
//   Timestamp -> Timestamp(us) (:208) | TimestampTz -> Timestamp(us, UTC) (:215) | Uuid -> FixedSizeBinary(16)
//   String -> LargeUtf8 | Bytes -> LargeBinary | numeric / json / arrays -> their heap entries (LargeBinary, the host finishes them)
//
// A cell the decode kernels handed back DEFERRED is null in `validity` and set in the column's `deferred` bitmap: the consumer
// finishes it from the arena (row_event names the event). Integer / byte work, HBM-bound: no MFMA.
#include "codec.hip.h"
#include "float_slow.h"
#include <type_traits>

namespace etlg {

DEV bool col_selected(const ColSel& s, uint64_t i, uint64_t& base) {
  if (i >= s.n_events || s.ev_slot[i] != s.slot) return false;
  const uint32_t k = s.ev_kind[i], fl = s.ev_flags[i];
  base = s.ev_body[i];
  if (k == 'I') return (s.kinds & 1u) != 0;
  if (k == 'U' && (s.kinds & 2u) && !(fl & ETLG_FLAG_PARTIAL)) {
    const uint32_t ok = fl & 3u;
    base += ok == ETLG_OLD_FULL ? s.row_full : ok == ETLG_OLD_KEY ? s.row_key : 0u;
    return true;
  }
  if (k == 'D' && (s.kinds & 4u) && (fl & 3u) == ETLG_OLD_FULL) return true;
  if (k == 'D' && (s.kinds & 8u) && (fl & 3u) == ETLG_OLD_KEY) return true;   // the key row: RowBinary expands it into the tombstone (rb_row)
  return false;
}

// ---- BigQuery: which rows an event becomes (bigquery/core.rs:978-1036, 1425-1476, 1557-1645)
// Did the update change the primary key (bigquery_primary_key_changed)? 0 no, 1 yes, 2 a cell the device does not compare.
DEV uint32_t pb_pk_changed(const ColSel& s, uint64_t oldb, uint64_t newb, bool key) {
  for (uint32_t i = 0; i < s.n_cols; i++) {
    const uint32_t kc = s.kcols[i];
    if (!(kc & 4u)) continue;
    const uint32_t cd = s.cols[i], cls = cd & 0xFFu;
    const uint32_t oi = key ? (kc >> 8) & 0xFFu : i, ooff = key ? kc >> 16 : cd >> 16;
    const uint32_t so = (s.fixed[oldb + oi / 4] >> (2 * (oi % 4))) & 3u, sn = (s.fixed[newb + i / 4] >> (2 * (i % 4))) & 3u;
    if ((so != ETLG_CELL_NULL && so != ETLG_CELL_VALUE) || (sn != ETLG_CELL_NULL && sn != ETLG_CELL_VALUE)) return 2;
    if (so != sn) return 1;
    if (so == ETLG_CELL_NULL) continue;
    const u8* a = s.fixed + oldb + ooff; const u8* b = s.fixed + newb + (cd >> 16);
    if (cls == ETLG_TC_STRING || cls == ETLG_TC_BYTEA) {
      const uint32_t la = *(const uint32_t*)(a + 4), lb = *(const uint32_t*)(b + 4);
      if (la != lb) return 1;
      const u8* ha = s.heap + *(const uint32_t*)a; const u8* hb = s.heap + *(const uint32_t*)b;
      for (uint32_t k = 0; k < la; k++) if (ha[k] != hb[k]) return 1;
    } else {
      const uint32_t nw = slot_bytes(cls) >> 2;
      for (uint32_t w = 0; w < nw; w++) if (((const uint32_t*)a)[w] != ((const uint32_t*)b)[w]) return 1;
    }
  }
  return 0;
}
// -> number of rows (0: the event stays with the host — the reference refuses it, or the device cannot decide), their bases
DEV uint32_t pb_selected(const ColSel& s, uint64_t i, unsigned long long* bases) {
  if (i >= s.n_events || s.ev_slot[i] != s.slot) return 0;
  const uint32_t k = s.ev_kind[i], fl = s.ev_flags[i], ok = fl & 3u;
  const uint64_t base = s.ev_body[i];
  if (k == 'I') { bases[0] = base; return 1; }
  if (k == 'D') {  // bigquery_delete_row: the old image's primary-key cells (a key image only under primary-key identity, :1540-1555)
    if (ok == ETLG_OLD_FULL) { bases[0] = base | kPbDelete; return 1; }
    if (ok == ETLG_OLD_KEY && s.identity_pk) { bases[0] = base | kPbDelete | kPbKey; return 1; }
    return 0;
  }
  if (k != 'U' || (fl & ETLG_FLAG_PARTIAL)) return 0;   // bigquery_update_new_row refuses partial rows (:1478-1494)
  const uint64_t newb = base + (ok == ETLG_OLD_FULL ? s.row_full : ok == ETLG_OLD_KEY ? s.row_key : 0u);
  if (ok == ETLG_OLD_NONE) {  // ensure_bigquery_update_without_old_row_can_skip_delete (:1515-1536)
    if (!s.identity_pk) return 0;
    bases[0] = newb; return 1;
  }
  if ((ok == ETLG_OLD_KEY && !s.identity_pk) || !s.pk_comparable) return 0;
  const uint32_t ch = pb_pk_changed(s, base, newb, ok == ETLG_OLD_KEY);
  if (ch == 2) return 0;
  if (ch == 1) {  // the old key goes first, the new row follows with the next ordinal (:1446-1474)
    bases[0] = base | kPbDelete | (ok == ETLG_OLD_KEY ? kPbKey : 0ull);
    bases[1] = newb | kPbSecond;
    return 2;
  }
  bases[0] = newb;
  return 1;
}

__global__ __launch_bounds__(256) void k_col_count(ColSel s) {
  __shared__ uint32_t lds[8];
  uint64_t base;
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned long long pbb[2];
  const uint32_t sel = s.pb ? pb_selected(s, i, pbb) : col_selected(s, i, base) ? 1u : 0u;
  if (s.host_rows) {  // row events of the slot that are not handed off
    bool left = false;
    if (!sel && i < s.n_events && s.ev_slot[i] == s.slot) { const uint32_t k = s.ev_kind[i]; left = k == 'I' || k == 'U' || k == 'D'; }
    const unsigned long long m = __ballot(left);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(s.host_rows, (unsigned long long)__builtin_popcountll(m));
  }
  uint32_t tot;
  block_scan_incl<0>(sel, lds, &tot);
  if (threadIdx.x == 0) s.blk[blockIdx.x] = tot;
}

// exclusive scan of n u32 counts in place (single workgroup, chunks of 256); total at [n]
__global__ __launch_bounds__(256) void k_col_scan(uint32_t* v, uint32_t n) {
  __shared__ uint32_t lds[8];
  uint32_t run = 0;
  for (uint32_t b0 = 0; b0 < n; b0 += 256) {
    const uint32_t i = b0 + threadIdx.x;
    const uint32_t x = i < n ? v[i] : 0u;
    uint32_t tot;
    const uint32_t inc = block_scan_incl<0>(x, lds, &tot);
    if (i < n) v[i] = run + inc - x;
    run += tot;
  }
  if (threadIdx.x == 0) v[n] = run;
}

__global__ __launch_bounds__(256) void k_col_rows(ColSel s) {
  __shared__ uint32_t lds[8];
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint64_t base = 0;
  unsigned long long pbb[2] = {0, 0};
  const uint32_t sel = s.pb ? pb_selected(s, i, pbb) : col_selected(s, i, base) ? 1u : 0u;
  const uint32_t inc = block_scan_incl<0>(sel, lds, nullptr);
  if (sel) {
    const uint32_t r = s.blk[blockIdx.x] + inc - sel;
    s.row_event[r] = i; s.row_base[r] = s.pb ? pbb[0] : base;
    if (sel == 2) { s.row_event[r + 1] = i; s.row_base[r + 1] = pbb[1]; }
  }
}

enum : uint32_t { AK_BOOL = 0, AK_I32 = 1, AK_I64 = 2, AK_F32 = 3, AK_F64 = 4, AK_DATE32 = 5, AK_TIME64 = 6, AK_TS = 7, AK_TSTZ = 8, AK_FIXED16 = 9,
                  AK_UTF8 = 10, AK_BINARY = 11, AK_TEXT_FORM = 12, AK_NUMERIC_STR = 14, AK_TIMETZ_STR = 15, AK_JSON_STR = 16, AK_NONE = 255 };   // 14 / 15 / 16: internal (host.cpp ColPlan.fmt), LargeUtf8 to the caller
constexpr int32_t kCeDays1970 = 719163;  // chrono num_days_from_ce of 1970-01-01

DEV uint32_t col_state(const ColJob& j, uint64_t base) { return (j.fixed[base + j.col_index / 4] >> (2 * (j.col_index % 4))) & 3u; }
DEV uint32_t ld32a(const u8* p) { return *(const uint32_t*)p; }   // row slots are 4-byte aligned

// One thread per row: state, value, validity / deferred words through wave ballots.
DEV void col_fixed_body(const ColJob& j, uint32_t bx) {
  const uint64_t r = (uint64_t)bx * 256 + threadIdx.x;
  const bool live = r < j.n_rows;
  uint32_t st = ETLG_CELL_NULL;
  const u8* slot = nullptr;
  if (live) { const uint64_t b = j.row_base[r]; st = col_state(j, b); slot = j.fixed + b + j.off_full; }
  const bool valid = live && st == ETLG_CELL_VALUE, defer = live && st == ETLG_CELL_DEFERRED;
  const unsigned long long vm = __ballot(valid), dm = __ballot(defer), lm = __ballot(live);
  if ((threadIdx.x & 63) == 0 && lm) {
    j.validity[r >> 6] = vm; j.deferred[r >> 6] = dm;
    const uint32_t nulls = (uint32_t)__builtin_popcountll(lm & ~vm), nd = (uint32_t)__builtin_popcountll(dm);
    if (nulls) atomicAdd(j.null_count, (unsigned long long)nulls);
    if (nd) atomicAdd(j.deferred_count, (unsigned long long)nd);
  }
  if (j.kind == AK_BOOL) {  // values are bit-packed like the validity
    const unsigned long long bits = __ballot(valid && ld32a(slot) != 0);
    if ((threadIdx.x & 63) == 0 && lm) ((unsigned long long*)j.values)[r >> 6] = bits;
    return;
  }
  if (!live) return;
  uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
  if (valid) { w0 = ld32a(slot); if (j.kind != AK_I32 && j.kind != AK_F32 && j.kind != AK_DATE32) w1 = ld32a(slot + 4); }
  switch (j.kind) {
    case AK_I32: case AK_F32: ((uint32_t*)j.values)[r] = w0; break;
    case AK_DATE32: ((int32_t*)j.values)[r] = valid ? (int32_t)w0 - kCeDays1970 : 0; break;
    case AK_I64: {  // I64 as is; U32 widens (cell_to_i64)
      const uint64_t v = j.cls == ETLG_TC_U32 ? (uint64_t)w0 : ((uint64_t)w1 << 32) | w0;
      ((uint64_t*)j.values)[r] = v; break;
    }
    case AK_F64: ((uint64_t*)j.values)[r] = ((uint64_t)w1 << 32) | w0; break;
    case AK_TIME64: ((int64_t*)j.values)[r] = valid ? (int64_t)w0 * 1000000 + (int64_t)(w1 / 1000u) : 0; break;
    case AK_TS: case AK_TSTZ: {
      if (valid) w2 = ld32a(slot + 8);
      const int64_t days = (int64_t)(int32_t)w0 - kCeDays1970;
      ((int64_t*)j.values)[r] = valid ? (days * 86400 + (int64_t)w1) * 1000000 + (int64_t)(w2 / 1000u) : 0; break;
    }
    case AK_FIXED16: {
      if (valid) { w2 = ld32a(slot + 8); w3 = ld32a(slot + 12); }
      ((uint4*)j.values)[r] = make_uint4(w0, w1, w2, w3); break;
    }
    default: break;
  }
}

__global__ __launch_bounds__(256) void k_col_fixed(ColJob j) { col_fixed_body(j, blockIdx.x); }

// Several columns of one hand-off in ONE launch (blockIdx.y = column): a cfg3 batch's Arrow columns were ~30 launches of 3-30 us each,
// and a third of the call was the gaps between them. The jobs travel in the kernel's argument block (a ColJob is 152 bytes, the
// block holds 4 KB): tables of more columns take several packs.
constexpr int kPack = 20;
struct ColPack { ColJob j[kPack]; unsigned long long* blk[kPack]; int64_t* offs[kPack]; unsigned long long* tot[kPack]; };   // tot: where the column's byte total goes (one read-back for all)
__global__ __launch_bounds__(256) void k_col_fixed_pack(ColPack p) { col_fixed_body(p.j[blockIdx.y], blockIdx.x); }

DEV int arr_hexv(uint32_t c) { return c - '0' < 10u ? (int)(c - '0') : (c | 0x20u) - 'a' < 6u ? (int)((c | 0x20u) - 'a' + 10) : -1; }
// serde_json 1.0.149 `from_str::<Value>` (call site codec/text.rs:126-134; features arbitrary_precision + std, crates/etl/Cargo.toml:36):
// is the text one JSON value? RFC 8259 grammar; whitespace is space / tab / LF / CR; a number keeps its literal text, so any length
// and exponent is fine; strings reject raw control characters, unknown escapes, a \u surrogate without its partner; an array or
// object may be nested 127 deep (Deserializer::remaining_depth starts at 128 and entering a container that takes it to 0 is
// RecursionLimitExceeded); anything but whitespace behind the value is an error. The text is valid UTF-8 already (the decode
// kernels checked). Iterative: the open containers are a 128-bit stack (1 = object).
DEV bool json_valid(const u8* s, uint32_t n) {
  uint32_t stk[4] = {0, 0, 0, 0};
  uint32_t depth = 0, i = 0;
  enum : uint32_t { X_VALUE = 0, X_VALUE_OR_CLOSE = 1, X_KEY_OR_CLOSE = 2, X_KEY = 3, X_NEXT = 4 };
  uint32_t ex = X_VALUE;
  auto ws = [&]() { while (i < n && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) i++; };
  auto hex4 = [&](uint32_t& v) -> bool {   // at s[i]: 'u' XXXX
    if (n - i < 5) return false;
    v = 0;
    for (uint32_t k = 1; k <= 4; k++) { const int d = arr_hexv(s[i + k]); if (d < 0) return false; v = v * 16 + (uint32_t)d; }
    i += 5;
    return true;
  };
  auto string = [&]() -> bool {   // at the opening quote
    i++;
    while (i < n) {
      const uint32_t c = s[i];
      if (c == '"') { i++; return true; }
      if (c < 0x20) return false;
      if (c != '\\') { i++; continue; }
      if (++i >= n) return false;
      const uint32_t x = s[i];
      if (x == 'u') {
        uint32_t v, w;
        if (!hex4(v)) return false;
        if (v >= 0xDC00 && v <= 0xDFFF) return false;
        if (v >= 0xD800 && v <= 0xDBFF) {
          if (n - i < 2 || s[i] != '\\' || s[i + 1] != 'u') return false;
          i++;
          if (!hex4(w) || w < 0xDC00 || w > 0xDFFF) return false;
        }
        continue;
      }
      if (!(x == '"' || x == '\\' || x == '/' || x == 'b' || x == 'f' || x == 'n' || x == 'r' || x == 't')) return false;
      i++;
    }
    return false;
  };
  auto digits = [&]() -> bool { const uint32_t i0 = i; while (i < n && s[i] - '0' < 10u) i++; return i > i0; };
  auto word = [&](const char* l, uint32_t ln) -> bool { if (n - i < ln) return false; for (uint32_t k = 0; k < ln; k++) if (s[i + k] != (u8)l[k]) return false; i += ln; return true; };
  for (;;) {
    ws();
    if (ex == X_NEXT) {
      if (!depth) return i == n;
      if (i >= n) return false;
      const uint32_t c = s[i++], top = depth - 1;
      const bool obj = (stk[top >> 5] >> (top & 31)) & 1u;
      if (c == ',') { ex = obj ? X_KEY : X_VALUE; continue; }
      if (c != (obj ? '}' : ']')) return false;
      depth--;
      continue;
    }
    if (i >= n) return false;
    const uint32_t c = s[i];
    if (ex == X_KEY_OR_CLOSE || ex == X_KEY) {
      if (ex == X_KEY_OR_CLOSE && c == '}') { i++; depth--; ex = X_NEXT; continue; }
      if (c != '"' || !string()) return false;
      ws();
      if (i >= n || s[i] != ':') return false;
      i++;
      ex = X_VALUE;
      continue;
    }
    if (ex == X_VALUE_OR_CLOSE && c == ']') { i++; depth--; ex = X_NEXT; continue; }
    if (c == '{' || c == '[') {
      if (depth >= 127) return false;
      if (c == '{') stk[depth >> 5] |= 1u << (depth & 31); else stk[depth >> 5] &= ~(1u << (depth & 31));
      depth++; i++;
      ex = c == '{' ? X_KEY_OR_CLOSE : X_VALUE_OR_CLOSE;
      continue;
    }
    bool ok;
    if (c == '"') ok = string();
    else if (c == 't') ok = word("true", 4);
    else if (c == 'f') ok = word("false", 5);
    else if (c == 'n') ok = word("null", 4);
    else {   // -? (0 | [1-9][0-9]*) (. [0-9]+)? ([eE] [+-]? [0-9]+)?
      if (c == '-') i++;
      if (i >= n) return false;
      if (s[i] == '0') i++; else if (s[i] - '1' < 9u) (void)digits(); else return false;
      ok = true;
      if (i < n && s[i] == '.') { i++; ok = digits(); }
      if (ok && i < n && (s[i] == 'e' || s[i] == 'E')) { i++; if (i < n && (s[i] == '+' || s[i] == '-')) i++; ok = digits(); }
    }
    if (!ok) return false;
    ex = X_NEXT;
  }
}


// ---- serde_json 1.0.149 `Value::to_string()` of a json / jsonb cell (the sinks' `j.to_string()`: clickhouse/encoding.rs:73,
// bigquery/encoding.rs:173-176, iceberg/encoding.rs:356) from the cell's source text, which json_valid() has accepted. What the parse +
// Display round trip changes (features arbitrary_precision + std, no preserve_order: crates/etl/Cargo.toml:36):
//   * whitespace between tokens goes; the output is compact ("," and ":" without blanks);
//   * an object is a BTreeMap<String, Value>: members leave in the byte order of their DECODED keys, a repeated key keeps its last value;
//   * strings are decoded and written again with serde_json's escapes: \" \\ \b \f \n \r \t, \u00xx (lowercase) for the other bytes
//     below 0x20, everything else raw (so "é" -> the two UTF-8 bytes, "\/" -> "/", a surrogate pair -> four bytes, 0x7f raw);
//   * a number keeps its literal (arbitrary_precision), except: an exponent without a sign gets '+' ("1e309" -> "1e+309": pinned by
//     the reference's own test, codec/text.rs:812-815), and an integer literal that fits u64 / i64 goes through the integer and back
//     (parse_any_number tries buf.parse() first), which only changes "-0" -> "0" (restated from serde_json's source; unpinned).
// Sorting is by selection: one pass over the object's members per member written (keys compared as decoded byte streams, no copy), so an
// object of k members costs k scans of its text. Returns 0; JD_HOST when the cell is beyond what a lane does here (nesting deeper than
// kJsonDepth, an object of more than kJsonMembers members, serde_json's private number token as a key) — the caller hands the cell
// back, as before; JD_BQ_INT when `bq` is set and a number that is WRITTEN (the value of a repeated key that lost is not in the parsed
// Value either) is an integer literal outside u64 / i64 (validate_json_number_for_bigquery, bigquery/validation.rs:64-85).
constexpr uint32_t kJsonDepth = 16, kJsonMembers = 64;
enum : uint32_t { JD_OK = 0, JD_HOST = 1, JD_BQ_INT = 2 };
struct JsIter { uint32_t i, pend, npend; };   // a cursor over a string's decoded bytes (i: behind the opening quote)
DEV int js_next(const u8* s, JsIter& k) {     // the next decoded byte, -1 at the closing quote
  if (k.npend) { const int b = (int)(k.pend & 0xFFu); k.pend >>= 8; k.npend--; return b; }
  const uint32_t c = s[k.i];
  if (c == '"') return -1;
  if (c != '\\') { k.i++; return (int)c; }
  const uint32_t x = s[k.i + 1];
  if (x != 'u') {
    k.i += 2;
    return x == 'b' ? 8 : x == 'f' ? 12 : x == 'n' ? 10 : x == 'r' ? 13 : x == 't' ? 9 : (int)x;   // \" \\ \/ are themselves
  }
  auto h4 = [&](uint32_t at) { uint32_t v = 0; for (uint32_t q = 0; q < 4; q++) v = v * 16 + (uint32_t)arr_hexv(s[at + q]); return v; };
  uint32_t cp = h4(k.i + 2);
  k.i += 6;
  if (cp >= 0xD800 && cp <= 0xDBFF) { cp = 0x10000 + ((cp - 0xD800) << 10) + (h4(k.i + 2) - 0xDC00); k.i += 6; }
  if (cp < 0x80) return (int)cp;
  if (cp < 0x800) { k.pend = 0x80 | (cp & 63); k.npend = 1; return (int)(0xC0 | (cp >> 6)); }
  if (cp < 0x10000) { k.pend = (0x80 | ((cp >> 6) & 63)) | ((0x80 | (cp & 63)) << 8); k.npend = 2; return (int)(0xE0 | (cp >> 12)); }
  k.pend = (0x80 | ((cp >> 12) & 63)) | ((0x80 | ((cp >> 6) & 63)) << 8) | ((0x80 | (cp & 63)) << 16); k.npend = 3;
  return (int)(0xF0 | (cp >> 18));
}
DEV int js_cmp(const u8* s, uint32_t a, uint32_t b) {   // the decoded strings at the opening quotes a and b: <0, 0, >0
  JsIter x{a + 1, 0, 0}, y{b + 1, 0, 0};
  for (;;) {
    const int p = js_next(s, x), q = js_next(s, y);
    if (p != q) return p - q;      // (-1, the end, sorts first: a prefix is smaller)
    if (p < 0) return 0;
  }
}
DEV uint32_t js_skip_string(const u8* s, uint32_t i) {   // from the opening quote to behind the closing one
  for (i++;; i++) { if (s[i] == '"') return i + 1; if (s[i] == '\\') i++; }
}
DEV uint32_t js_ws(const u8* s, uint32_t i, uint32_t n) { while (i < n && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) i++; return i; }
DEV uint32_t js_skip_value(const u8* s, uint32_t i, uint32_t n) {   // from a value's first byte to behind it
  const uint32_t c = s[i];
  if (c == '"') return js_skip_string(s, i);
  if (c == '{' || c == '[') {
    uint32_t d = 0;
    for (;;) {
      const uint32_t x = s[i];
      if (x == '"') { i = js_skip_string(s, i); continue; }
      if (x == '{' || x == '[') d++;
      else if (x == '}' || x == ']') { if (!--d) return i + 1; }
      i++;
    }
  }
  while (i < n && s[i] != ',' && s[i] != '}' && s[i] != ']' && s[i] != ' ' && s[i] != '\t' && s[i] != '\n' && s[i] != '\r') i++;
  return i;
}
template <class S>
DEV uint32_t js_put_string(S& out, const u8* s, uint32_t i) {   // the string at the opening quote i, escaped again; returns behind it
  JsIter k{i + 1, 0, 0};
  out.put('"');
  for (;;) {
    const int b = js_next(s, k);
    if (b < 0) break;
    if (b == '"' || b == '\\') { out.put('\\'); out.put((u8)b); }
    else if (b >= 0x20) out.put((u8)b);
    else {
      out.put('\\');
      if (b == 8) out.put('b'); else if (b == 12) out.put('f'); else if (b == 10) out.put('n'); else if (b == 13) out.put('r'); else if (b == 9) out.put('t');
      else { out.put('u'); out.put('0'); out.put('0'); out.put((u8)('0' + (b >> 4))); out.put((u8)((b & 15) < 10 ? '0' + (b & 15) : 'a' + (b & 15) - 10)); }
    }
  }
  out.put('"');
  return k.i + 1;
}
template <class S>
DEV uint32_t json_display(S& out, const u8* s, uint32_t n, bool bq) {
  uint32_t f_start[kJsonDepth], f_last[kJsonDepth], f_end[kJsonDepth];   // objects: behind '{', the last key written, behind '}'
  uint32_t is_obj = 0, depth = 0;
  uint32_t i = js_ws(s, 0, n);
  constexpr uint32_t NONE = 0xFFFFFFFFu;
  for (;;) {
    // ---- one value at i
    bool opened = false;
    {
      const uint32_t c = s[i];
      if (c == '"') i = js_put_string(out, s, i);
      else if (c == '{' || c == '[') {
        if (depth >= kJsonDepth) return JD_HOST;
        out.put((u8)c);
        if (c == '{') { is_obj |= 1u << depth; f_start[depth] = i + 1; f_last[depth] = NONE; f_end[depth] = 0; }
        else {
          is_obj &= ~(1u << depth);
          i = js_ws(s, i + 1, n);
          if (s[i] == ']') { out.put(']'); i++; goto after_value; }   // (depth not raised: an empty array is a value like any other)
          opened = true;
        }
        depth++;
        if (opened) continue;   // the array's first element sits at i
      } else if (c == 't' || c == 'f' || c == 'n') { const uint32_t e = js_skip_value(s, i, n); for (; i < e; i++) out.put(s[i]); }
      else {   // a number
        const uint32_t e = js_skip_value(s, i, n);
        bool integer = true;
        for (uint32_t q = i; q < e; q++) if (s[q] == '.' || s[q] == 'e' || s[q] == 'E') integer = false;
        if (integer && bq) {   // number.parse::<i64>() / ::<u64>() must succeed
          const bool neg = s[i] == '-';
          const uint32_t d0 = i + (neg ? 1 : 0), nd = e - d0;
          const char* lim = neg ? "9223372036854775808" : "18446744073709551615";
          const uint32_t nl = neg ? 19 : 20;
          bool over = nd > nl;
          if (nd == nl) { for (uint32_t q = 0; q < nl; q++) { if (s[d0 + q] != (u8)lim[q]) { over = s[d0 + q] > (u8)lim[q]; break; } } }
          if (over) return JD_BQ_INT;
        }
        if (e - i == 2 && s[i] == '-' && s[i + 1] == '0') { out.put('0'); i = e; }
        else for (; i < e; i++) { out.put(s[i]); if ((s[i] == 'e' || s[i] == 'E') && s[i + 1] != '+' && s[i + 1] != '-') out.put('+'); }
      }
    }
    // ---- behind a value (or inside a fresh object): what the innermost open container wants next
  after_value:
    for (;;) {
      if (!depth) return JD_OK;
      const uint32_t t = depth - 1;
      if (!((is_obj >> t) & 1u)) {   // array: the text goes on in order
        i = js_ws(s, i, n);
        if (s[i] == ',') { out.put(','); i = js_ws(s, i + 1, n); break; }
        out.put(']'); i++; depth--;
        continue;
      }
      // object: the smallest key above the last one written; of equal keys the last
      uint32_t p = js_ws(s, f_start[t], n), best = NONE, bestv = 0, members = 0;
      const uint32_t last = f_last[t];
      while (s[p] != '}') {
        if (s[p] == ',') p = js_ws(s, p + 1, n);
        const uint32_t kq = p;
        p = js_ws(s, js_skip_string(s, p), n) + 1;   // behind ':'
        p = js_ws(s, p, n);
        const uint32_t v = p;
        p = js_ws(s, js_skip_value(s, p, n), n);
        if (++members > kJsonMembers) return JD_HOST;
        if (last == NONE) {   // the first round also looks for the private token ("$serde_json::private::Number" as a key makes from_str read a number)
          const char* tok = "$serde_json::private::";
          JsIter it{kq + 1, 0, 0};
          bool is_tok = true;
          for (uint32_t q = 0; q < 22 && is_tok; q++) is_tok = js_next(s, it) == (int)tok[q];
          if (is_tok) return JD_HOST;
        }
        if (last != NONE && js_cmp(s, kq, last) <= 0) continue;
        if (best == NONE || js_cmp(s, kq, best) <= 0) { best = kq; bestv = v; }
      }
      f_end[t] = p + 1;
      if (best == NONE) { out.put('}'); i = f_end[t]; depth--; continue; }
      if (last != NONE) out.put(',');
      f_last[t] = best;
      (void)js_put_string(out, s, best);
      out.put(':');
      i = bestv;
      break;
    }
  }
}
struct JsCount { uint32_t n = 0; DEV void put(u8) { n++; } };

DEV uint32_t numeric_str_len(const u8* ent);
DEV uint32_t timetz_str_len(const u8* slot);
// var-len columns, pass 1: validity / deferred words + the byte length of every row's entry
// (blk: the block's sum of lengths, for the offsets scan — a launch of its own, k_col_len_blocks, for the callers that have no such pass)
// JS: the launch has a json column that leaves as its Display string (a kernel of its own: json_display's registers would cost every
// other table two waves per SIMD)
template <bool JS>
DEV void col_lens_body(const ColJob& j, unsigned long long* blk, uint32_t bx, uint64_t* lds_sum) {
  const uint64_t r = (uint64_t)bx * 256 + threadIdx.x;
  const bool live = r < j.n_rows;
  uint32_t st = ETLG_CELL_NULL, len = 0;
  if (live) {
    const uint64_t b = j.row_base[r];
    st = col_state(j, b);
    // text-form columns hand over DEFERRED entries too (their heap entry is the source text)
    const bool text_form = j.kind == AK_TEXT_FORM || j.kind == AK_JSON_STR;
    const bool has = st == ETLG_CELL_VALUE || (text_form && st == ETLG_CELL_DEFERRED);
    bool json_ok = true;
    if (has && j.cls == ETLG_TC_JSON) {   // "JSON deserialization failed" for the first malformed cell in event order (codec/text.rs:126-134)
      const u8* slot = j.fixed + b + j.off_full;
      json_ok = json_valid(j.heap + ld32a(slot), ld32a(slot + 4));
      if (!json_ok) atomicMin(j.err, (unsigned long long)((r << 8) | ETLG_E_JSON));
    }
    if (has) {
      const u8* slot = j.fixed + b + j.off_full;
      len = j.kind == AK_NUMERIC_STR ? numeric_str_len(j.heap + ld32a(slot)) : j.kind == AK_TIMETZ_STR ? timetz_str_len(slot) : ld32a(slot + 4);
      if (JS && j.kind == AK_JSON_STR && json_ok) {   // the Display string where a lane writes it, the source text handed back DEFERRED where not
        JsCount c;
        if (json_display(c, j.heap + ld32a(slot), len, false) == JD_OK) { len = c.n; st = ETLG_CELL_VALUE; }
      }
    }
    j.lens[r] = len;
  }
  const bool valid = live && (st == ETLG_CELL_VALUE || ((j.kind == AK_TEXT_FORM || j.kind == AK_JSON_STR) && st == ETLG_CELL_DEFERRED));
  const bool defer = live && st == ETLG_CELL_DEFERRED;
  const unsigned long long vm = __ballot(valid), dm = __ballot(defer), lm = __ballot(live);
  if ((threadIdx.x & 63) == 0 && lm) {
    j.validity[r >> 6] = vm; j.deferred[r >> 6] = dm;
    const uint32_t nulls = (uint32_t)__builtin_popcountll(lm & ~vm), nd = (uint32_t)__builtin_popcountll(dm);
    if (nulls) atomicAdd(j.null_count, (unsigned long long)nulls);
    if (nd) atomicAdd(j.deferred_count, (unsigned long long)nd);
  }
  const uint64_t t = block_sum64(len, lds_sum);
  if (threadIdx.x == 0) blk[bx] = t;
}
template <bool JS>
__global__ __launch_bounds__(256) void k_col_lens(ColJob j, unsigned long long* blk) {
  __shared__ uint64_t lds_sum[4];
  col_lens_body<JS>(j, blk, blockIdx.x, lds_sum);
}
template <bool JS>
__global__ __launch_bounds__(256) void k_col_lens_pack(ColPack p) {
  __shared__ uint64_t lds_sum[4];
  col_lens_body<JS>(p.j[blockIdx.y], p.blk[blockIdx.y], blockIdx.x, lds_sum);
}

// lens (u32) -> offsets (i64), three steps like k_col_count / k_col_scan / k_col_rows
__global__ __launch_bounds__(256) void k_col_len_blocks(const uint32_t* lens, uint64_t n, unsigned long long* blk) {
  __shared__ uint64_t lds[4];
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t t = block_sum64(i < n ? lens[i] : 0u, lds);
  if (threadIdx.x == 0) blk[blockIdx.x] = t;
}
DEV void col_len_scan_body(unsigned long long* blk, uint32_t n, uint64_t* lds) {
  uint64_t run = 0;
  for (uint32_t b0 = 0; b0 < n; b0 += 256) {
    const uint32_t i = b0 + threadIdx.x;
    uint64_t tot;
    const uint64_t ex = block_scan_excl64(i < n ? blk[i] : 0ull, lds, &tot);
    if (i < n) blk[i] = run + ex;
    run += tot;
  }
  if (threadIdx.x == 0) blk[n] = run;
}
__global__ __launch_bounds__(256) void k_col_len_scan(unsigned long long* blk, uint32_t n) {
  __shared__ uint64_t lds[4];
  col_len_scan_body(blk, n, lds);
}
__global__ __launch_bounds__(256) void k_col_len_scan_pack(ColPack p, uint32_t n) {   // one workgroup per column
  __shared__ uint64_t lds[4];
  col_len_scan_body(p.blk[blockIdx.x], n, lds);
}
DEV void col_offsets_body(const uint32_t* lens, uint64_t n, const unsigned long long* blk, int64_t* offsets, uint32_t bx, uint64_t* lds, unsigned long long* tot = nullptr) {
  const uint64_t i = (uint64_t)bx * 256 + threadIdx.x;
  const uint64_t ex = block_scan_excl64(i < n ? lens[i] : 0u, lds, nullptr);
  if (i < n) offsets[i] = (int64_t)(blk[bx] + ex);
  if (i == n - 1) { offsets[n] = (int64_t)(blk[bx] + ex + lens[i]); if (tot) *tot = blk[bx] + ex + lens[i]; }
}
__global__ __launch_bounds__(256) void k_col_offsets(const uint32_t* lens, uint64_t n, const unsigned long long* blk, int64_t* offsets, unsigned long long* tot = nullptr) {
  __shared__ uint64_t lds[4];
  col_offsets_body(lens, n, blk, offsets, blockIdx.x, lds, tot);
}
__global__ __launch_bounds__(256) void k_col_offsets_pack(ColPack p) {
  __shared__ uint64_t lds[4];
  const ColJob& j = p.j[blockIdx.y];
  col_offsets_body(j.lens, j.n_rows, p.blk[blockIdx.y], p.offs[blockIdx.y], blockIdx.x, lds, p.tot[blockIdx.y]);
}

// var-len columns, pass 2: one wave per 64 rows; the wave moves one row at a time, 4 bytes per lane per step where both ends
// allow it (heap entries start 4-byte aligned; the destination is wherever the previous row ended)
DEV void col_copy_body(const ColJob& j, uint32_t bx) {
  // A wave takes 64 consecutive rows (their bytes are consecutive in `values`): eight lanes per row, eight rows at a time, eight bytes
  // per lane and step. (One row at a time with a byte per lane was a load and a store instruction per row of up to 64 bytes: 44 us per
  // text column of a cfg3 batch, profiles/r04q.)
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t r0 = ((uint64_t)bx * 4 + (threadIdx.x >> 6)) * 64;
  if (r0 >= j.n_rows) return;
  const uint64_t r = r0 + lane;
  uint32_t len = 0, src = 0; int64_t dst = 0;
  if (r < j.n_rows) { len = j.lens[r]; dst = j.offsets[r]; if (len) src = ld32a(j.fixed + j.row_base[r] + j.off_full); }
  const uint32_t sub = lane & 7u, grp = lane >> 3;
  for (uint32_t it = 0; it < 8; it++) {
    const int k = (int)(it * 8u + grp);
    const uint32_t l_k = (uint32_t)__shfl((int)len, k, 64);
    const uint32_t s_k = (uint32_t)__shfl((int)src, k, 64);
    const uint64_t d_k = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)((uint64_t)dst >> 32), k, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)dst, k, 64);
    const u8* sp = j.heap + s_k;
    u8* dp = j.values + d_k;
    for (uint32_t b = sub * 8u; b < l_k; b += 64u) {
      if (b + 8u <= l_k) { uint64_t v; __builtin_memcpy(&v, sp + b, 8); __builtin_memcpy(dp + b, &v, 8); }
      else for (uint32_t t = b; t < l_k; t++) dp[t] = sp[t];
    }
  }
}
__global__ __launch_bounds__(256) void k_col_copy(ColJob j) { col_copy_body(j, blockIdx.x); }


// ---- Display strings of the classes every sink writes as text: PgNumeric (format_numeric_value,
// crates/etl-postgres/src/numeric.rs:460-560), PgTimeTz (etl-postgres/src/time.rs:113-117 + write_utc_offset :210-225) and
// chrono's "%H:%M:%S%.f" (TIME_FORMAT, time.rs:17). `ent`: the numeric's heap entry (etlg_numeric_hdr + i16 digits, 4-byte aligned).
// All of them are written through the same count / write sinks as the rows (RbCount / RbWrite below), so a length is the
// count of the very code that later writes the bytes — except the numeric's, which has a closed form (a scale can be 16383).
DEV uint32_t num_digit(const u8* ent, uint32_t i) { return (uint32_t)ent[8 + 2 * i] | ((uint32_t)ent[9 + 2 * i] << 8); }
DEV uint32_t numeric_str_len(const u8* ent) {
  const uint32_t kind = ent[0];
  if (kind == ETLG_NUM_NAN) return 3;        // "NaN"
  if (kind == ETLG_NUM_PINF) return 8;       // "Infinity"
  if (kind == ETLG_NUM_NINF) return 9;       // "-Infinity"
  const int32_t weight = (int16_t)((uint32_t)ent[2] | ((uint32_t)ent[3] << 8));
  const uint32_t scale = (uint32_t)ent[4] | ((uint32_t)ent[5] << 8), nd = (uint32_t)ent[6] | ((uint32_t)ent[7] << 8);
  const uint32_t frac = scale ? 1u + scale : 0u;
  if (!nd) return 1u + frac;                 // zero keeps its display scale (:492-503)
  uint32_t n = ent[1] ? 1u : 0u;
  if (weight < 0) n += 1u;
  else { const uint32_t d0 = num_digit(ent, 0); n += (d0 >= 1000 ? 4u : d0 >= 100 ? 3u : d0 >= 10 ? 2u : 1u) + 4u * (uint32_t)weight; }
  return n + frac;
}
template <class S> DEV void put_4d(S& s, uint32_t v) { s.put((u8)('0' + v / 1000 % 10)); s.put((u8)('0' + v / 100 % 10)); s.put((u8)('0' + v / 10 % 10)); s.put((u8)('0' + v % 10)); }
template <class S>
DEV void numeric_str(S& s, const u8* ent) {
  const uint32_t kind = ent[0];
  if (kind != ETLG_NUM_VALUE) {
    const char* t = kind == ETLG_NUM_NAN ? "NaN" : kind == ETLG_NUM_PINF ? "Infinity" : "-Infinity";
    for (; *t; t++) s.put((u8)*t);
    return;
  }
  const int32_t weight = (int16_t)((uint32_t)ent[2] | ((uint32_t)ent[3] << 8));
  const uint32_t scale = (uint32_t)ent[4] | ((uint32_t)ent[5] << 8), nd = (uint32_t)ent[6] | ((uint32_t)ent[7] << 8);
  if (!nd) {
    s.put('0');
    if (scale) { s.put('.'); for (uint32_t k = 0; k < scale; k++) s.put('0'); }
    return;
  }
  if (ent[1]) s.put('-');
  if (weight < 0) s.put('0');
  else {
    for (int32_t d = 0; d <= weight; d++) {
      const uint32_t g = (uint32_t)d < nd ? num_digit(ent, (uint32_t)d) : 0u;
      if (d == 0) {  // the first group without its leading zeros (:517-524)
        if (g >= 1000) s.put((u8)('0' + g / 1000 % 10));
        if (g >= 100) s.put((u8)('0' + g / 100 % 10));
        if (g >= 10) s.put((u8)('0' + g / 10 % 10));
        s.put((u8)('0' + g % 10));
      } else put_4d(s, g);
    }
  }
  if (scale) {
    s.put('.');
    // `let mut d = weight + 1` is i16 arithmetic in the reference (:535): at weight = i16::MAX a release build wraps to
    // i16::MIN and prints zeros; restated as such
    int32_t d = (int16_t)(weight + 1);
    for (uint32_t rem = scale; rem; d++) {
      const uint32_t g = (d >= 0 && (uint32_t)d < nd) ? num_digit(ent, (uint32_t)d) : 0u;
      const uint32_t take = rem < 4 ? rem : 4u;
      uint32_t div = 1000;
      for (uint32_t k = 0; k < take; k++, div /= 10) s.put((u8)('0' + g / div % 10));
      rem -= take;
    }
  }
}
template <class S> DEV void put_2d(S& s, uint32_t v) { s.put((u8)('0' + v / 10)); s.put((u8)('0' + v % 10)); }
// chrono's %.f prints nothing, or 3 / 6 / 9 digits; a leap second is kept as nanos >= 10^9 on second 59 and printed as :60
DEV uint32_t time_frac_len(uint32_t nanos) { nanos = nanos >= 1000000000u ? nanos - 1000000000u : nanos; return nanos == 0 ? 0u : nanos % 1000000u == 0 ? 4u : nanos % 1000u == 0 ? 7u : 10u; }
template <class S> DEV void time_str(S& s, uint32_t secs, uint32_t nanos) {
  const uint32_t leap = nanos >= 1000000000u ? 1u : 0u;
  nanos -= leap * 1000000000u;
  put_2d(s, secs / 3600); s.put(':'); put_2d(s, secs / 60 % 60); s.put(':'); put_2d(s, secs % 60 + leap);
  const uint32_t frac = time_frac_len(nanos);
  if (frac) {
    s.put('.');
    uint32_t v = frac == 4 ? nanos / 1000000u : frac == 7 ? nanos / 1000u : nanos, div = frac == 4 ? 100u : frac == 7 ? 100000u : 100000000u;
    for (; div; div /= 10) s.put((u8)('0' + v / div % 10));
  }
}
DEV uint32_t utc_offset_len(int32_t off) { const uint32_t a = (uint32_t)(off < 0 ? -off : off); return a % 60 ? 9u : a % 3600 ? 6u : 3u; }
template <class S> DEV void utc_offset_str(S& s, int32_t off) {   // +HH | +HH:MM | +HH:MM:SS (write_utc_offset)
  const uint32_t a = (uint32_t)(off < 0 ? -off : off);
  s.put(off < 0 ? '-' : '+');
  put_2d(s, a / 3600);
  if (a % 60) { s.put(':'); put_2d(s, a % 3600 / 60); s.put(':'); put_2d(s, a % 60); }
  else if (a % 3600) { s.put(':'); put_2d(s, a % 3600 / 60); }
}
DEV uint32_t timetz_str_len(const u8* slot) { return 8u + time_frac_len(ld32a(slot + 4)) + utc_offset_len((int32_t)ld32a(slot + 8)); }
template <class S> DEV void timetz_str(S& s, const u8* slot) { time_str(s, ld32a(slot), ld32a(slot + 4)); utc_offset_str(s, (int32_t)ld32a(slot + 8)); }

// The byte pass of a row. One thread writes one row, so a byte store per put() was one write request per BYTE at the L2 (64 lanes,
// 64 different lines per instruction): k_rb_rows took 469 us for the 47 MB of a cfg3 batch's rows (profiles/r04q). The bytes are
// collected in a 64-bit accumulator instead and leave eight at a time (unaligned 8-byte stores are fine in global memory); finish()
// writes the last 1-7 bytes one by one — the next row's first bytes belong to another thread.
struct RbGlobalSink {
  u8* p;                // where the accumulator's first byte goes
  uint64_t acc = 0;
  uint32_t n = 0;       // bytes in acc (0..7)
  DEV explicit RbGlobalSink(u8* q) : p(q) {}
  DEV void store8(uint64_t v) { __builtin_memcpy(p, &v, 8); p += 8; }
  // appends the low k bytes of v (1 <= k <= 8; the bytes above them are zero)
  DEV void append(uint64_t v, uint32_t k) {
    acc |= v << (8u * n);
    const uint32_t m = n + k;
    if (m >= 8u) {
      store8(acc);
      acc = n ? v >> (8u * (8u - n)) : 0ull;   // what did not fit (n = 0: k = 8, nothing is left)
      n = m - 8u;
    } else n = m;
  }
  DEV void finish() { for (uint32_t b = 0; b < n; b++) p[b] = (u8)(acc >> (8u * b)); p += n; n = 0; acc = 0; }
};
// The same bytes into a ZEROED image of the output in LDS (k_rb_rows): whole words are OR-ed in (ds_or_b32), so the first and the last
// word of a part may be shared with its neighbours; the image leaves for global memory in 16-byte stores of the whole workgroup.
struct RbLdsSink {
  uint32_t* w;          // the word the accumulator's first byte belongs to
  uint64_t acc = 0;
  uint32_t n;           // bytes in acc (0..3 between calls), counting the bytes of *w in front of this part
  DEV RbLdsSink(uint32_t* word, uint32_t lead) : w(word), n(lead) {}
  DEV void app4(uint32_t v, uint32_t k) {   // 1 <= k <= 4
    acc |= (uint64_t)v << (8u * n);
    n += k;
    if (n >= 4u) { atomicOr(w++, (uint32_t)acc); acc >>= 32; n -= 4u; }
  }
  DEV void append(uint64_t v, uint32_t k) { if (k > 4u) { app4((uint32_t)v, 4u); app4((uint32_t)(v >> 32), k - 4u); } else app4((uint32_t)v, k); }
  DEV void finish() { if (n && (uint32_t)acc) atomicOr(w, (uint32_t)acc); n = 0; acc = 0; }
};
template <class B>
struct RbWriterT : B {
  using B::B;
  using B::append;
  DEV void put(u8 b) { append(b, 1); }
  DEV void varint64(uint64_t v) { while (v >= 0x80) { put((u8)(v | 0x80)); v >>= 7; } put((u8)v); }
  DEV void put32(uint32_t v) { append(v, 4); }
  DEV void put64(uint64_t v) { append(v, 8); }
  DEV void zeros(uint32_t k) { while (k >= 8u) { append(0ull, 8); k -= 8u; } if (k) append(0ull, k); }
  DEV void bytes(const u8* s, uint32_t len) {
    uint32_t k = 0;
    for (; k + 16u <= len; k += 16u) { uint64_t v[2]; __builtin_memcpy(v, s + k, 16); append(v[0], 8); append(v[1], 8); }   // (one request per 16 bytes of a long text)
    for (; k + 8u <= len; k += 8u) { uint64_t v; __builtin_memcpy(&v, s + k, 8); append(v, 8); }
    if (k < len) { uint64_t v = 0; for (uint32_t b = 0; k + b < len; b++) v |= (uint64_t)s[k + b] << (8u * b); append(v, len - k); }
  }
  DEV void hex(const u8* s, uint32_t len) {   // bytes_to_hex, lowercase (:176-185)
    auto h1 = [](uint32_t d) -> uint64_t { return d < 10 ? '0' + d : 'a' + d - 10; };
    uint32_t k = 0;
    for (; k + 4u <= len; k += 4u) {   // four bytes -> eight digits
      uint64_t v = 0;
      for (uint32_t b = 0; b < 4; b++) { const uint32_t x = s[k + b]; v |= (h1(x >> 4) | (h1(x & 15u) << 8)) << (16u * b); }
      append(v, 8);
    }
    for (; k < len; k++) { const uint32_t x = s[k]; append(h1(x >> 4) | (h1(x & 15u) << 8), 2); }
  }
};
using RbWrite = RbWriterT<RbGlobalSink>;
using RbLdsWrite = RbWriterT<RbLdsSink>;
// formatted string columns (numeric, timetz), pass 2: one thread per row writes its Display string at its offset
template <bool JS>
DEV void col_fmt_body(const ColJob& j, uint32_t bx) {
  const uint64_t r = (uint64_t)bx * 256 + threadIdx.x;
  if (r >= j.n_rows || !j.lens[r]) return;
  const u8* slot = j.fixed + j.row_base[r] + j.off_full;
  // through the row formats' writer: eight bytes per store (a store per byte was a write request per byte and lane). For json the byte
  // writer did worse than that: its instantiation faulted on the MI355X — a store address with its low or high half replaced — while
  // the same function under a bounds-checked byte writer, under RbWrite in k_rb_rows and on the emulator was right
  // (profiles/r05x_json_arrow_fault.txt; not pursued further).
  RbWrite w(j.values + j.offsets[r]);
  if (j.kind == AK_NUMERIC_STR) numeric_str(w, j.heap + ld32a(slot));
  else if (j.kind == AK_TIMETZ_STR) timetz_str(w, slot);
  else if (JS) {   // json: the Display string, or (the rows pass 1 marked DEFERRED) the source text as it is
    const u8* t = j.heap + ld32a(slot);
    if ((j.deferred[r >> 6] >> (r & 63)) & 1ull) w.bytes(t, j.lens[r]);
    else (void)json_display(w, t, ld32a(slot + 4), false);
  }
  w.finish();
}
template <bool JS>
__global__ __launch_bounds__(256) void k_col_fmt(ColJob j) { col_fmt_body<JS>(j, blockIdx.x); }
// pass 2 of several var-len columns in one launch: a column is copied or formatted by what it is (uniform per blockIdx.y)
template <bool JS>
__global__ __launch_bounds__(256) void k_col_var2_pack(ColPack p) {
  const ColJob& j = p.j[blockIdx.y];
  if (j.kind == AK_NUMERIC_STR || j.kind == AK_TIMETZ_STR || j.kind == AK_JSON_STR) col_fmt_body<JS>(j, blockIdx.x); else col_copy_body(j, blockIdx.x);
}

// ---- array literals (parse_cell_from_postgres_text_array, crates/etl/src/postgres/codec/text.rs:228-312; the dimensions
// prefix :163-214) for the element classes with a fixed-width value: bool, int2, int4, int8, oid, float4, float8, date, time,
// timestamp, timestamptz, uuid. One thread per row walks its
// text twice: k_arr_count (shape errors, element parse errors, element count), then k_arr_fill behind the offsets scan.
constexpr uint32_t kArrElemMax = 40;   // an element text longer than this is left to the host (Rust accepts any number of leading zeros)
enum : uint32_t { ARR_HOST = 0x100 };  // not an error: the row is handed back deferred

DEV uint32_t arr_strip_dims(const u8* s, uint32_t n, uint32_t& start) {   // strip_array_dimensions_prefix
  auto at = [&](uint32_t i) -> int { return i < n ? (int)s[i] : -1; };
  start = 0;
  if (at(0) != '[') return 0;
  uint32_t groups = 0, idx = 0;
  auto skip_int = [&](uint32_t i, uint32_t& out) { if (at(i) == '-') i++; const uint32_t st = i; while (at(i) >= '0' && at(i) <= '9') i++; out = i; return i > st; };
  while (at(idx) == '[') {
    uint32_t a, b;
    if (!skip_int(idx + 1, a) || at(a) != ':') return ETLG_E_ARRAY_DIMS;
    if (!skip_int(a + 1, b) || at(b) != ']') return ETLG_E_ARRAY_DIMS;
    idx = b + 1; groups++;
  }
  if (at(idx) != '=') return ETLG_E_ARRAY_DIMS;
  if (groups > 1) return ETLG_E_ARRAY_MULTIDIM;
  start = idx + 1;
  return 0;
}

// Calls elem(k, is_null, value words) per element in text order; returns 0, an etlg_err_code, or ARR_HOST.
// TEXT: string elements (ArrayCell::String: text[], varchar[], and every array type without a dedicated arm) are the unescaped
// bytes themselves; `dst(k)` says where element k's bytes go (nullptr: they are only counted).
// BYTEA elements (ArrayCell::Bytes, parse_bytea_hex_string per element, codec/hex.rs:11-52): TEXT walks with `hex` set — the element's
// characters are "\x" + hex pairs, decoded as they come; w[0] = the byte count.
template <bool TEXT, class F, class D>
DEV uint32_t arr_walk(const u8* s0, uint32_t n0, uint32_t elem_cls, uint32_t& count, F&& elem, D&& dst, bool exact_floats = false) {
  const bool hex = TEXT && elem_cls == ETLG_TC_BYTEA;
  bool hex_bad = false; uint32_t nib = 0;
  uint32_t start;
  count = 0;
  if (const uint32_t e = arr_strip_dims(s0, n0, start)) return e;
  const u8* s = s0 + start;
  const uint32_t n = n0 - start;
  if (n < 2) return ETLG_E_ARRAY_SHORT;
  if (s[0] != '{' || s[n - 1] != '}') return ETLG_E_ARRAY_BRACES;
  const u8* body = s + 1;
  const uint32_t bn = n - 2;
  u8 val[kArrElemMax];
  uint32_t vl = 0, pos = 0;
  bool in_quotes = false, in_escape = false, val_quoted = false, done = bn == 0, too_long = false;
  u8* out = TEXT ? dst(0u) : nullptr;
  while (!done) {
    for (;;) {
      if (pos >= bn) { done = true; break; }
      const u8 c = body[pos++];
      bool push = false;
      if (in_escape) { push = true; in_escape = false; }
      else if (c == '"') { if (!in_quotes) val_quoted = true; in_quotes = !in_quotes; }
      else if (c == '\\') in_escape = true;
      else if ((c == '{' || c == '}') && !in_quotes) return ETLG_E_ARRAY_MULTIDIM;
      else if (c == ',' && !in_quotes) break;
      else push = true;
      if (push) {
        if (vl < kArrElemMax) val[vl] = c; else too_long = true;
        if (hex) {   // characters 0, 1: "\x"; then pairs (an unquoted "null" has no backslash: it never looks like bytes)
          if (vl == 0) hex_bad |= c != '\\';
          else if (vl == 1) hex_bad |= c != 'x';
          else {
            const int h = arr_hexv(c);
            hex_bad |= h < 0;
            if (vl & 1) { if (out && !hex_bad) out[(vl - 3) >> 1] = (u8)((nib << 4) | (uint32_t)(h & 15)); } else nib = (uint32_t)(h & 15);   // (an unquoted NULL is four non-hex characters: it must not write)
          }
        }
        // a text element's bytes leave as they come, except the first four: an unquoted "null" is not text at all
        else if (TEXT && out && vl >= 4) { if (vl == 4) { out[0] = val[0]; out[1] = val[1]; out[2] = val[2]; out[3] = val[3]; } out[vl] = c; }
        vl++;
      }
    }
    if (in_quotes) return ETLG_E_ARRAY_QUOTE;
    if (in_escape) return ETLG_E_ARRAY_ESCAPE;
    if (!TEXT && too_long) return ARR_HOST;
    const bool is_null = !val_quoted && vl == 4 && (val[0] | 0x20) == 'n' && (val[1] | 0x20) == 'u' && (val[2] | 0x20) == 'l' && (val[3] | 0x20) == 'l';
    uint32_t w[4] = {0, 0, 0, 0};
    uint32_t scratch[(kArrElemMax + 7) / 4 + 2];   // a numeric element's heap entry (header + digits of <= 40 characters); a DEFERRED element's text
    if (hex) {
      if (!is_null) {   // "Bytea hex string conversion failed": no "\x", an odd count, a non-hex character (hex.rs:21-50)
        if (vl < 2 || hex_bad || (vl & 1)) return ETLG_E_BYTEA;
        w[0] = (vl - 2) >> 1;
      }
      hex_bad = false;
    } else if (TEXT) {
      if (out && !is_null && vl <= 4) for (uint32_t b = 0; b < vl; b++) out[b] = val[b];
      w[0] = is_null ? 0u : vl;
    } else if (!is_null) {
      uint32_t hcur = 0, st = 0;
      if (const uint32_t e = decode_text_cell<true>(elem_cls, val, vl, w, (u8*)scratch, hcur, st, false)) return e;
      if (st != ETLG_CELL_VALUE) {   // a float text the fast rule does not settle: the exact conversion (finish pass), else the host's
        if (!exact_floats || !(elem_cls == ETLG_TC_F32 || elem_cls == ETLG_TC_F64)) return ARR_HOST;
        const uint64_t bits = parse_float_exact_t([&](uint32_t i) { return (uint32_t)val[i]; }, vl, elem_cls == ETLG_TC_F32);
        w[0] = (uint32_t)bits; w[1] = (uint32_t)(bits >> 32);
      }
    }
    elem(count, is_null, w, (const u8*)scratch);
    count++;
    vl = 0; val_quoted = false;
    if (TEXT) out = dst(count);
  }
  return 0;
}

// The same walk for the row formats' text-like elements (ArrayCell::String / Bytes), which need an element's LENGTH in front of its
// bytes: elem(k, is_null, p0, p1, ulen) gets the element's source characters s0[p0 .. p1) — quotes and backslashes included — and its
// unescaped length; arr_unescape() then replays the span. Same checks, same NULL rule (an unquoted, unescaped "null" of any case).
template <class F>
DEV uint32_t arr_spans(const u8* s0, uint32_t n0, uint32_t& count, F&& elem) {
  uint32_t start;
  count = 0;
  if (const uint32_t e = arr_strip_dims(s0, n0, start)) return e;
  const uint32_t n = n0 - start;
  if (n < 2) return ETLG_E_ARRAY_SHORT;
  if (s0[start] != '{' || s0[start + n - 1] != '}') return ETLG_E_ARRAY_BRACES;
  const uint32_t b0 = start + 1, b1 = start + n - 1;   // the body
  uint32_t pos = b0;
  bool done = b1 == b0;
  while (!done) {
    const uint32_t p0 = pos;
    uint32_t p1 = b1, vl = 0, low4 = 0;
    bool in_quotes = false, in_escape = false, val_quoted = false, escaped = false;
    for (;;) {
      if (pos >= b1) { done = true; p1 = b1; break; }
      const u8 c = s0[pos++];
      bool push = false;
      if (in_escape) { push = true; in_escape = false; }
      else if (c == '"') { if (!in_quotes) val_quoted = true; in_quotes = !in_quotes; }
      else if (c == '\\') { in_escape = true; escaped = true; }
      else if ((c == '{' || c == '}') && !in_quotes) return ETLG_E_ARRAY_MULTIDIM;
      else if (c == ',' && !in_quotes) { p1 = pos - 1; break; }
      else push = true;
      if (push) { if (vl < 4) low4 |= (uint32_t)(c | 0x20) << (8 * vl); vl++; }
    }
    if (in_quotes) return ETLG_E_ARRAY_QUOTE;
    if (in_escape) return ETLG_E_ARRAY_ESCAPE;
    const bool is_null = !val_quoted && vl == 4 && low4 == 0x6C6C756Eu;   // "null" (an escaped n\ull is "null" too: the reference compares the unescaped value)
    (void)escaped;
    elem(count, is_null, p0, p1, vl);
    count++;
  }
  return 0;
}
template <class E>
DEV void arr_unescape(const u8* s0, uint32_t p0, uint32_t p1, E&& emit) {
  bool esc = false;
  for (uint32_t p = p0; p < p1; p++) {
    const u8 c = s0[p];
    if (esc) { emit(c); esc = false; } else if (c == '\\') esc = true; else if (c != '"') emit(c);
  }
}
// a bytea element's unescaped text: "\x" + hex pairs (parse_bytea_hex_string, codec/hex.rs:11-52)? Returns the byte count, or ~0u.
DEV uint32_t arr_bytea_len(const u8* s0, uint32_t p0, uint32_t p1, uint32_t ulen) {
  if (ulen < 2 || (ulen & 1)) return ~0u;
  uint32_t k = 0; bool bad = false;
  arr_unescape(s0, p0, p1, [&](u8 c) { if (k == 0) bad |= c != '\\'; else if (k == 1) bad |= c != 'x'; else bad |= arr_hexv(c) < 0; k++; });
  return bad ? ~0u : (ulen - 2) >> 1;
}

DEV bool arr_text(const ColJob& j, uint64_t r, const u8*& s, uint32_t& n, uint32_t& st) {
  const uint64_t b = j.row_base[r];
  st = col_state(j, b);
  if (st != ETLG_CELL_VALUE && st != ETLG_CELL_DEFERRED) return false;
  const u8* slot = j.fixed + b + j.off_full;
  s = j.heap + ld32a(slot); n = ld32a(slot + 4);
  return true;
}

// json[] / jsonb[] as a list of `j.to_string()` strings (ArrayCell::Json, iceberg/encoding.rs:577, 964): the literal's elements one by
// one — unescaped into private memory (json_display walks its text back and forth), checked as ONE JSON value (an element that is not
// is the reference's decode error, codec/text.rs:126-134, like a scalar json cell), sized. An element of more than kJsonElemMax bytes
// or beyond json_display's limits (depth 16, 64 members) hands the row back (ARR_HOST). kJsonElemMax is defined with the row formats below.
constexpr uint32_t kJsonListElemMax = 256;
DEV uint32_t arr_json_check(const u8* s, uint32_t n, uint32_t& cnt) {
  u8 tmp[kJsonListElemMax];
  bool too_long = false, bad_json = false, limit = false;
  const uint32_t e = arr_spans(s, n, cnt, [&](uint32_t, bool is_null, uint32_t p0, uint32_t p1, uint32_t ulen) {
    if (is_null || too_long || bad_json) return;
    if (ulen > kJsonListElemMax) { too_long = true; return; }
    uint32_t k = 0;
    arr_unescape(s, p0, p1, [&](u8 c) { tmp[k++] = c; });
    if (!json_valid(tmp, ulen)) { bad_json = true; return; }
    JsCount c;
    if (json_display(c, tmp, ulen, false)) limit = true;
  });
  if (e) return e;
  if (too_long) return ARR_HOST;          // (an element too long to look at may not be JSON at all: the row is the host's before anything else)
  if (bad_json) return ETLG_E_JSON;
  return limit ? (uint32_t)ARR_HOST : 0u;
}

__global__ __launch_bounds__(256) void k_arr_count(ColJob j) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = r < j.n_rows;
  bool valid = false, defer = false;
  if (live) {
    const u8* s; uint32_t n, st, cnt = 0;
    if (arr_text(j, r, s, n, st)) {
      auto none = [](uint32_t) -> u8* { return nullptr; };
      auto skip = [](uint32_t, bool, const uint32_t*, const u8*) {};
      const uint32_t e = j.elem_cls == ETLG_TC_JSON ? arr_json_check(s, n, cnt)
                       : (j.elem_cls == ETLG_TC_STRING || j.elem_cls == ETLG_TC_BYTEA) ? arr_walk<true>(s, n, j.elem_cls, cnt, skip, none) : arr_walk<false>(s, n, j.elem_cls, cnt, skip, none);
      // A row handed back (ARR_HOST) is the host's to finish, and it may turn out to be the batch's first malformed literal. So the call
      // only fails for a malformed row when no handed-back row precedes it (the host compares the two minima: code 0xFF marks a
      // hand-back); otherwise the malformed rows are handed back as well, and the consumer — finishing deferred rows in event
      // order — meets the first problem first, like parse_cell_from_postgres_text at decode time.
      if (e == ARR_HOST) { defer = true; cnt = 0; atomicMin(j.err, (unsigned long long)((r << 8) | 0xFFu)); }
      else if (e) { atomicMin(j.err, (unsigned long long)((r << 8) | e)); cnt = 0; defer = true; }
      else valid = true;
    }
    j.lens[r] = cnt;
  }
  const unsigned long long vm = __ballot(valid), dm = __ballot(defer), lm = __ballot(live);
  if ((threadIdx.x & 63) == 0 && lm) {
    j.validity[r >> 6] = vm; j.deferred[r >> 6] = dm;
    const uint32_t nulls = (uint32_t)__builtin_popcountll(lm & ~vm), nd = (uint32_t)__builtin_popcountll(dm);
    if (nulls) atomicAdd(j.null_count, (unsigned long long)nulls);
    if (nd) atomicAdd(j.deferred_count, (unsigned long long)nd);
  }
}

__global__ __launch_bounds__(256) void k_arr_fill(ColJob j) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= j.n_rows || !j.lens[r]) return;
  const u8* s; uint32_t n, st, cnt;
  if (!arr_text(j, r, s, n, st)) return;
  const uint64_t o = (uint64_t)j.offsets[r];
  uint32_t nulls = 0;
  if (j.elem_cls == ETLG_TC_JSON) {
    // pass A (values not set): the Display length and validity of every element; pass B: the characters
    u8 tmp[kJsonListElemMax];
    if (!j.values) {
      (void)arr_spans(s, n, cnt, [&](uint32_t k, bool is_null, uint32_t p0, uint32_t p1, uint32_t ulen) {
        const uint64_t e = o + k;
        uint32_t len = 0;
        if (!is_null) {
          uint32_t q = 0;
          arr_unescape(s, p0, p1, [&](u8 c) { tmp[q++] = c; });
          JsCount c;
          (void)json_display(c, tmp, ulen, false);
          len = c.n;
          atomicOr(&j.child_validity[e >> 5], 1u << (e & 31));
        } else nulls++;
        j.child_lens[e] = len;
      });
      if (nulls) atomicAdd(j.child_nulls, (unsigned long long)nulls);
    } else {
      (void)arr_spans(s, n, cnt, [&](uint32_t k, bool is_null, uint32_t p0, uint32_t p1, uint32_t ulen) {
        if (is_null) return;
        uint32_t q = 0;
        arr_unescape(s, p0, p1, [&](u8 c) { tmp[q++] = c; });
        RbWrite sw(j.values + j.child_offsets[o + k]);
        (void)json_display(sw, tmp, ulen, false);
        sw.finish();
      });
    }
    return;
  }
  if (j.elem_cls == ETLG_TC_NUMERIC || j.elem_cls == ETLG_TC_TIMETZ) {
    // ArrayCell::Numeric / TimeTz are lists of their Display strings in the sinks (iceberg/encoding.rs:902-945: `n.to_string()`,
    // `t.to_string()`): pass A their lengths, pass B the characters. The element's value is what decode_text_cell left in the
    // walker's scratch (a numeric heap entry) or in its slot words (timetz).
    const bool num = j.elem_cls == ETLG_TC_NUMERIC;
    if (!j.values) {
      (void)arr_walk<false>(s, n, j.elem_cls, cnt, [&](uint32_t k, bool is_null, const uint32_t* w, const u8* scratch) {
        const uint64_t e = o + k;
        j.child_lens[e] = is_null ? 0u : num ? numeric_str_len(scratch + w[0]) : timetz_str_len((const u8*)w);
        if (is_null) nulls++; else atomicOr(&j.child_validity[e >> 5], 1u << (e & 31));
      }, [](uint32_t) -> u8* { return nullptr; });
      if (nulls) atomicAdd(j.child_nulls, (unsigned long long)nulls);
    } else {
      (void)arr_walk<false>(s, n, j.elem_cls, cnt, [&](uint32_t k, bool is_null, const uint32_t* w, const u8* scratch) {
        if (is_null) return;
        RbWrite sw(j.values + j.child_offsets[o + k]);
        if (num) numeric_str(sw, scratch + w[0]); else timetz_str(sw, (const u8*)w);
        sw.finish();
      }, [](uint32_t) -> u8* { return nullptr; });
    }
    return;
  }
  if (j.elem_cls == ETLG_TC_STRING || j.elem_cls == ETLG_TC_BYTEA) {
    // pass A (child_lens set, values not): the byte length and validity of every element; pass B (values set): the bytes
    if (!j.values) {
      (void)arr_walk<true>(s, n, j.elem_cls, cnt, [&](uint32_t k, bool is_null, const uint32_t* w, const u8*) {
        const uint64_t e = o + k;
        j.child_lens[e] = w[0];
        if (is_null) nulls++; else atomicOr(&j.child_validity[e >> 5], 1u << (e & 31));
      }, [](uint32_t) -> u8* { return nullptr; });
      if (nulls) atomicAdd(j.child_nulls, (unsigned long long)nulls);
    } else {
      (void)arr_walk<true>(s, n, j.elem_cls, cnt, [](uint32_t, bool, const uint32_t*, const u8*) {},
                           [&](uint32_t k) -> u8* { return k < j.lens[r] ? j.values + j.child_offsets[o + k] : nullptr; });
    }
    return;
  }
  (void)arr_walk<false>(s, n, j.elem_cls, cnt, [&](uint32_t k, bool is_null, const uint32_t* w, const u8*) {
    const uint64_t e = o + k;
    if (is_null) { nulls++; }
    else atomicOr(&j.child_validity[e >> 5], 1u << (e & 31));
    switch (j.kind) {   // child layout: the same conversions as k_col_fixed
      case AK_BOOL: if (!is_null && w[0]) atomicOr(&((uint32_t*)j.values)[e >> 5], 1u << (e & 31)); break;
      case AK_I32: case AK_F32: ((uint32_t*)j.values)[e] = w[0]; break;
      case AK_DATE32: ((int32_t*)j.values)[e] = is_null ? 0 : (int32_t)w[0] - kCeDays1970; break;
      case AK_TIME64: ((int64_t*)j.values)[e] = is_null ? 0 : (int64_t)w[0] * 1000000 + (int64_t)(w[1] / 1000u); break;
      case AK_TS: case AK_TSTZ:
        ((int64_t*)j.values)[e] = is_null ? 0 : (((int64_t)(int32_t)w[0] - kCeDays1970) * 86400 + (int64_t)w[1]) * 1000000 + (int64_t)(w[2] / 1000u); break;
      case AK_FIXED16: ((uint4*)j.values)[e] = make_uint4(w[0], w[1], w[2], w[3]); break;
      default: ((uint64_t*)j.values)[e] = j.elem_cls == ETLG_TC_U32 ? (uint64_t)w[0] : ((uint64_t)w[1] << 32) | w[0]; break;   // I64, U32, F64
    }
  }, [](uint32_t) -> u8* { return nullptr; });
  if (nulls) atomicAdd(j.child_nulls, (unsigned long long)nulls);
}

// ---- ClickHouse RowBinary (crates/etl-destinations/src/clickhouse/encoding.rs:58-83 which wire type a Cell becomes,
// :188-283 the byte format; core.rs:96-114 the trailing CDC columns). One thread per row, run twice: lengths, then bytes.
enum : uint32_t { RB_E_NULL = 1, RB_E_DATE_RANGE = 2, RB_E_HOST_CELL = 3, RB_E_BQ_NUMERIC_SCALE = 4 /* and the json integer rule: one report */, RB_E_JSON = 5, RB_E_BQ_ARRAY_NULL = 6 };
constexpr int32_t kDate32Min = -25567, kDate32Max = 120529;   // 1900-01-01 .. 2299-12-31 (encoding.rs:147-173)

struct RbCount {
  uint32_t n = 0;
  DEV void put(u8) { n++; }
  DEV void varint64(uint64_t v) { do { n++; v >>= 7; } while (v); }
  DEV void put32(uint32_t) { n += 4; }
  DEV void put64(uint64_t) { n += 8; }
  DEV void zeros(uint32_t k) { n += k; }
  DEV void bytes(const u8*, uint32_t len) { n += len; }
  DEV void hex(const u8*, uint32_t len) { n += 2 * len; }
};

template <class S>
DEV void rb_varint(S& s, uint32_t v) {   // LEB128 (:188-199)
  while (v >= 0x80) { s.put((u8)(v | 0x80)); v >>= 7; }
  s.put((u8)v);
}

template <class S>
DEV void rb_2d(S& s, uint32_t v) { put_2d(s, v); }

constexpr uint32_t kJsonElemMax = 256;   // a json[] element longer than this (unescaped) is left to the host: a lane unescapes it into private memory
// A json cell as the sinks' `j.to_string()`: `head(len)` writes what goes in front of the string (its varint length). The text is
// checked in the counting pass only (a row that fails has length 0 and is not written). Returns 0, RB_E_JSON (not one JSON value: the
// reference fails at decode time, codec/text.rs:126-134), RB_E_HOST_CELL (json_display leaves it to the host), RB_E_BQ_NUMERIC_SCALE.
template <class S, class H>
DEV uint32_t rb_json(S& s, const u8* t, uint32_t tn, bool bq, H head) {
  if (std::is_same<S, RbCount>::value && !json_valid(t, tn)) return RB_E_JSON;
  JsCount c;
  const uint32_t e = json_display(c, t, tn, bq);
  if (e) return e == JD_BQ_INT ? RB_E_BQ_NUMERIC_SCALE : RB_E_HOST_CELL;
  head(c.n);
  if (std::is_same<S, RbCount>::value) s.zeros(c.n); else (void)json_display(s, t, tn, false);
  return 0;
}

// One non-null value of class `cls` whose arena slot words start at `slot` (a row's slot, or the words decode_text_cell produced for
// an array element). Returns 0, RB_E_DATE_RANGE or RB_E_HOST_CELL.
// The counting pass notes where the row's `qparts` pieces (4, or 2 / 1 for narrow tables) begin; the byte pass writes a row with
// parts <= qparts lanes (1, 2 or 4 — chosen when the bytes per row are known), a lane taking qparts / parts pieces.
DEV uint32_t rb_part_col(const RbJob& j, uint32_t q) { return q * j.n_cols / j.qparts; }   // the first column of piece q

template <bool JS = false, class S>
DEV uint32_t rb_scalar(S& s, uint32_t cls, const u8* slot, const u8* heap) {
  const uint32_t w0 = ld32a(slot);
  switch (cls) {
    case ETLG_TC_BOOL: s.put(w0 ? 1 : 0); return 0;
    case ETLG_TC_I16: s.put((u8)w0); s.put((u8)(w0 >> 8)); return 0;
    case ETLG_TC_I32: case ETLG_TC_U32: case ETLG_TC_F32: s.put32(w0); return 0;
    case ETLG_TC_I64: case ETLG_TC_F64: s.put64(((uint64_t)ld32a(slot + 4) << 32) | w0); return 0;
    case ETLG_TC_DATE: {
      const int32_t days = (int32_t)w0 - kCeDays1970;
      if (days < kDate32Min || days > kDate32Max) return RB_E_DATE_RANGE;
      s.put32((uint32_t)days); return 0;
    }
    case ETLG_TC_TIME: {  // String(t.to_string()): chrono NaiveTime Display
      const uint32_t nanos = ld32a(slot + 4);
      rb_varint(s, 8 + time_frac_len(nanos)); time_str(s, w0, nanos);
      return 0;
    }
    case ETLG_TC_TIMETZ: rb_varint(s, timetz_str_len(slot)); timetz_str(s, slot); return 0;   // String(t.to_string()) (encoding.rs:71)
    case ETLG_TC_NUMERIC: if (!heap) return RB_E_HOST_CELL; { const u8* ent = heap + w0; rb_varint(s, numeric_str_len(ent)); numeric_str(s, ent); return 0; }   // String(n.to_string()) (:66)
    case ETLG_TC_TIMESTAMP: case ETLG_TC_TIMESTAMPTZ: {
      const int64_t days = (int64_t)(int32_t)w0 - kCeDays1970;
      s.put64((uint64_t)((days * 86400 + (int64_t)ld32a(slot + 4)) * 1000000 + (int64_t)(ld32a(slot + 8) / 1000u))); return 0;
    }
    case ETLG_TC_UUID:  // high u64 LE then low u64 LE of the big-endian 16 bytes (:240-247)
      for (int h = 0; h < 2; h++) for (int k = 7; k >= 0; k--) s.put(slot[8 * h + k]);
      return 0;
    case ETLG_TC_STRING: if (!heap) return RB_E_HOST_CELL; { const uint32_t len = ld32a(slot + 4); rb_varint(s, len); s.bytes(heap + w0, len); return 0; }
    case ETLG_TC_BYTEA: if (!heap) return RB_E_HOST_CELL; { const uint32_t len = ld32a(slot + 4); rb_varint(s, 2 * len); s.hex(heap + w0, len); return 0; }
    case ETLG_TC_JSON: if (!JS || !heap) return RB_E_HOST_CELL; return rb_json(s, heap + w0, ld32a(slot + 4), false, [&](uint32_t len) { rb_varint(s, len); });   // String(j.to_string()) (:73)
    default: return RB_E_HOST_CELL;
  }
}

// default_cell(typ) (clickhouse/core.rs:1481-1517) in RowBinary is a run of zero bytes: the type's width for the fixed-width classes
// (false, 0, 0.0, Date32 day 0 = 1970-01-01, DateTime64 0 = the epoch, Uuid::nil()), the varint 0 of an empty Array / an empty String
// (numeric, time, timetz, interval, bytea, text ...) for the rest. One length + one zero run instead of a switch over put32 / put64 /
// put64 x 2: hipcc (ROCm 7.2) compiled that switch inside rb_row's column loop with the output pointer left undefined behind the
// TIMESTAMP / TIMESTAMPTZ arm (k_rb_rows wrote through a stale register on the MI355X; profiles/r04_rowbinary_tombstone_fault.txt).
DEV uint32_t rb_default_zero_bytes(uint32_t cls) {
  uint32_t n = 1;                                                                                          // BOOL, and every String / Array class
  if (cls == ETLG_TC_I16) n = 2;
  if (cls == ETLG_TC_I32 || cls == ETLG_TC_U32 || cls == ETLG_TC_F32 || cls == ETLG_TC_DATE) n = 4;
  if (cls == ETLG_TC_I64 || cls == ETLG_TC_F64 || cls == ETLG_TC_TIMESTAMP || cls == ETLG_TC_TIMESTAMPTZ) n = 8;
  if (cls == ETLG_TC_UUID) n = 16;
  return n;
}

// [c_lo, c_hi): the columns this call writes (the byte pass splits a row among several lanes, k_rb_rows; the trailing columns go with
// the last part); mark(i) is called in front of column i (the counting pass notes where the parts begin).
template <bool JS, class S, class M>
DEV uint32_t rb_row(const RbJob& j, uint64_t r, S& s, uint32_t c_lo, uint32_t c_hi, M&& mark) {   // returns 0, or column << 8 | code of the cell that fails the row
  // The reference converts every cell of every pending row first (cell_to_clickhouse_value, clickhouse/core.rs:1193-1203: Date32
  // range errors) and only then encodes the rows (NULL in a non-nullable column): a range error anywhere beats a NULL error. So a
  // row that meets a cell without an encoding goes on looking for a date out of range; k_rb_lens ranks range errors first across rows.
  // A json cell that is not one JSON value fails earlier still — at decode time in the reference — so it beats both.
  const uint64_t base = j.row_base[r];
  uint32_t err0 = 0, errd = 0;
  // A Delete that carries only the key becomes the tombstone expand_key_row builds (clickhouse/core.rs:1437-1472): the key cells in
  // the primary-key columns, NULL in every other column that is nullable at the source and not an array, default_cell's zero value
  // (:1481-1517) in the rest. The host selects such rows only where the reference accepts them (host_handoff.inc).
  const bool keyrow = j.kcols && j.ev_kind[j.row_event[r]] == 'D' && (j.ev_flags[j.row_event[r]] & 3u) == ETLG_OLD_KEY;
  for (uint32_t i = c_lo; i < c_hi; i++) {
    mark(i);
    const uint32_t cd = j.cols[i], cls = cd & 0xFF;
    uint32_t off = cd >> 16, sti = i;
    const bool nullable = (cd >> 8) & 1;
    if (keyrow) {
      const uint32_t kc = j.kcols[i];
      if (kc & 1u) { off = kc >> 16; sti = (kc >> 8) & 0xFFu; }   // an identity column: its cell sits in the key layout
      else if ((kc & 2u) && cls != ETLG_TC_ARRAY) {                // Cell::Null
        if (!nullable) { if (!err0) err0 = (i << 8) | RB_E_NULL; continue; }
        s.put(1);
        continue;
      } else {                                                    // default_cell(typ)
        if (nullable) s.put(0);
        s.zeros(rb_default_zero_bytes(cls));
        continue;
      }
    }
    const uint32_t st = (j.fixed[base + sti / 4] >> (2 * (sti % 4))) & 3u;
    if (st == ETLG_CELL_NULL) {
      if (!nullable) { if (!err0) err0 = (i << 8) | RB_E_NULL; continue; }   // "NULL value for non-nullable ClickHouse column" (:217-225)
      s.put(1);
      continue;
    }
    const u8* slot = j.fixed + base + off;
    if (cls == ETLG_TC_ARRAY && st != ETLG_CELL_MISSING) {
      // Array(Nullable(T)) (:249-254): varint count, then every element with its null marker. The literal (kept as text in the
      // arena) is walked twice: count, then encode. A literal the device cannot take apart is the host's (it raises the exact error).
      const uint32_t elem = (cd >> 9) & 0x7Fu;
      if (elem == ETLG_TC_JSON && !JS) { if (!err0) err0 = (i << 8) | RB_E_HOST_CELL; continue; }
      const u8* txt = j.heap + ld32a(slot);
      const uint32_t tn = ld32a(slot + 4);
      if (JS && elem == ETLG_TC_JSON) {
        // json[] / jsonb[]: String(j.to_string()) per element (encoding.rs:109). An element is unescaped into private memory (json_display
        // walks its text back and forth); one that is not JSON is the reference's decode error, as for a scalar cell.
        u8 tmp[kJsonElemMax];
        uint32_t cnt = 0, bad = 0;
        bool too_long = false, bad_json = false, limit = false;   // (an element too long to look at may not be JSON at all: the cell is the host's before anything else)
        if (arr_spans(txt, tn, cnt, [&](uint32_t, bool is_null, uint32_t p0, uint32_t p1, uint32_t ulen) {
              if (is_null) return;
              if (ulen > kJsonElemMax) { too_long = true; return; }
              uint32_t k = 0;
              arr_unescape(txt, p0, p1, [&](u8 c) { tmp[k++] = c; });
              if (std::is_same<S, RbCount>::value && !json_valid(tmp, ulen)) { bad_json = true; return; }
              JsCount c;
              if (json_display(c, tmp, ulen, false)) limit = true;
            })) too_long = true;
        bad = too_long ? RB_E_HOST_CELL : bad_json ? RB_E_JSON : limit ? RB_E_HOST_CELL : 0u;
        if (bad == RB_E_JSON) return (i << 8) | bad;
        if (bad) { if (!err0) err0 = (i << 8) | bad; continue; }
        if (nullable) s.put(0);
        rb_varint(s, cnt);
        (void)arr_spans(txt, tn, cnt, [&](uint32_t, bool is_null, uint32_t p0, uint32_t p1, uint32_t ulen) {
          if (is_null) { s.put(1); return; }
          s.put(0);
          uint32_t k = 0;
          arr_unescape(txt, p0, p1, [&](u8 c) { tmp[k++] = c; });
          JsCount c;
          (void)json_display(c, tmp, ulen, false);
          rb_varint(s, c.n);
          if (std::is_same<S, RbCount>::value) s.zeros(c.n); else (void)json_display(s, tmp, ulen, false);
        });
        continue;
      }
      if (elem == ETLG_TC_STRING || elem == ETLG_TC_BYTEA) {
        // text-like elements are String(the unescaped bytes), bytea elements String(bytes_to_hex(..)) (array_cell_to_clickhouse_values,
        // encoding.rs:89-111): the lowercase hex digits of the element's own "\x.." text
        uint32_t cnt = 0; bool bad = false;
        if (arr_spans(txt, tn, cnt, [&](uint32_t, bool is_null, uint32_t p0, uint32_t p1, uint32_t ulen) {
              if (elem == ETLG_TC_BYTEA && !is_null && arr_bytea_len(txt, p0, p1, ulen) == ~0u) bad = true;
            }) || bad) { if (!err0) err0 = (i << 8) | RB_E_HOST_CELL; continue; }
        if (nullable) s.put(0);
        rb_varint(s, cnt);
        (void)arr_spans(txt, tn, cnt, [&](uint32_t, bool is_null, uint32_t p0, uint32_t p1, uint32_t ulen) {
          if (is_null) { s.put(1); return; }
          s.put(0);
          if (elem == ETLG_TC_STRING) { rb_varint(s, ulen); arr_unescape(txt, p0, p1, [&](u8 c) { s.put(c); }); }
          else { uint32_t k = 0; rb_varint(s, ulen - 2); arr_unescape(txt, p0, p1, [&](u8 c) { if (k++ >= 2) s.put((u8)(c - 'A' < 6u ? c | 0x20 : c)); }); }
        });
        continue;
      }
      uint32_t cnt = 0;
      auto none = [](uint32_t) -> u8* { return nullptr; };
      if (arr_walk<false>(txt, tn, elem, cnt, [](uint32_t, bool, const uint32_t*, const u8*) {}, none)) { if (!err0) err0 = (i << 8) | RB_E_HOST_CELL; continue; }
      if (nullable) s.put(0);
      rb_varint(s, cnt);
      uint32_t ee = 0;
      (void)arr_walk<false>(txt, tn, elem, cnt, [&](uint32_t, bool is_null, const uint32_t* w, const u8* scratch) {
        if (is_null) { s.put(1); return; }
        s.put(0);
        const uint32_t e1 = rb_scalar(s, elem, (const u8*)w, scratch);   // (a numeric element's entry sits in the walk's scratch: String(n.to_string()); timetz: String(t.to_string()))
        if (e1 && !ee) ee = e1;
      }, none);
      if (ee == RB_E_DATE_RANGE) { if (!errd) errd = (i << 8) | ee; }
      else if (ee && !err0) err0 = (i << 8) | ee;
      continue;
    }
    if (st != ETLG_CELL_VALUE && !(cls == ETLG_TC_JSON && st == ETLG_CELL_DEFERRED)) { if (!err0) err0 = (i << 8) | RB_E_HOST_CELL; continue; }   // (json cells are source text in the arena: DEFERRED)
    if (nullable) s.put(0);
    if (const uint32_t e = rb_scalar<JS>(s, cls, slot, j.heap)) {
      if (e == RB_E_JSON) return (i << 8) | e;
      if (e == RB_E_DATE_RANGE) { if (!errd) errd = (i << 8) | e; }
      else if (!err0) err0 = (i << 8) | e;
    }
  }
  if (errd) return errd;
  if (err0) return err0;
  if (c_hi != j.n_cols) return 0;
  // trailing CDC columns (core.rs:96-114); never NULL, a Nullable() destination column still takes its marker byte
  const uint64_t ev = j.row_event[r];
  const uint32_t kind = j.ev_kind[ev];
  const uint64_t lsn = j.ev_commit[ev], ord = j.ev_ord[ev];
  if (j.cdc_nullable & 1u) s.put(0);
  if (j.engine == 0) {
    s.put(6);
    const char* op = kind == 'I' ? "INSERT" : kind == 'U' ? "UPDATE" : "DELETE";
    for (int k = 0; k < 6; k++) s.put((u8)op[k]);
    if (j.cdc_nullable & 2u) s.put(0);
    s.put64(lsn);
  } else {
    s.put64(ord); s.put64(lsn);   // u128 = commit_lsn << 64 | tx_ordinal, little endian
    if (j.cdc_nullable & 2u) s.put(0);
    s.put(kind == 'D' ? 1 : 0);
  }
  return 0;
}


// ---- BigQuery protobuf rows (cell_encode_prost, crates/etl-destinations/src/bigquery/encoding.rs:120-190; the wire format is
// prost's = protobuf's: key = varint(tag << 3 | wire type), wire types 0 varint, 1 fixed64, 2 length-delimited, 5 fixed32; int32 /
// int64 as sign-extended 64-bit varints). Insert rows only: the row's cells with tags 1..n (NULL cells leave nothing), then
// _CHANGE_TYPE = "UPSERT" and _CHANGE_SEQUENCE_NUMBER = "{commit_lsn:016x}/{tx_ordinal:016x}/{0:016x}" (bigquery/core.rs:978-996,
// 1404-1406; EventSequenceKey Display, crates/etl/src/event.rs:346-351).
template <class S> DEV void pb_key(S& s, uint32_t tag, uint32_t wt) { s.varint64(((uint64_t)tag << 3) | wt); }
template <class S> DEV void pb_4d(S& s, uint32_t v) { s.put((u8)('0' + v / 1000 % 10)); s.put((u8)('0' + v / 100 % 10)); s.put((u8)('0' + v / 10 % 10)); s.put((u8)('0' + v % 10)); }
template <class S> DEV void pb_hex16(S& s, uint64_t v) { for (int k = 15; k >= 0; k--) { const uint32_t d = (uint32_t)(v >> (4 * k)) & 15u; s.put((u8)(d < 10 ? '0' + d : 'a' + d - 10)); } }
// days from CE (chrono) -> civil date: "%Y-%m-%d" (DATE_FORMAT, etl-postgres/src/time.rs:13); years 0000-9999 (the others are DEFERRED)
template <class S> DEV void pb_date(S& s, int32_t days_ce) {
  const int64_t z = (int64_t)days_ce - kCeDays1970 + 719468;
  const int64_t era = (z >= 0 ? z : z - 146096) / 146097;
  const uint32_t doe = (uint32_t)(z - era * 146097);
  const uint32_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  const uint32_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  const uint32_t mp = (5 * doy + 2) / 153;
  const uint32_t d = doy - (153 * mp + 2) / 5 + 1, m = mp < 10 ? mp + 3 : mp - 9;
  const int64_t y = (int64_t)yoe + era * 400 + (m <= 2 ? 1 : 0);
  pb_4d(s, (uint32_t)y); s.put('-'); rb_2d(s, m); s.put('-'); rb_2d(s, d);
}
// An array cell (array_cell_encode_prost, bigquery/encoding.rs:203-290) behind validate_array_cell_for_bigquery (validation.rs:125-190:
// a NULL element fails the row). Element classes with a fixed-width value, as in RowBinary: bool / int2 / int4 / oid / int8 / float4 /
// float8 and timestamptz (epoch microseconds) leave PACKED — one length-delimited field of the values back to back (varints, or 4- / 8-
// byte words), nothing at all for an empty array; date / time / timestamp / uuid leave as one string field per element. The literal is
// walked once for the element count, the NULLs and the packed length, once for the bytes. A literal the device does not take apart
// (malformed: the reference's decode error, which the host raises; an element of more than 40 characters) is RB_E_HOST_CELL.
template <bool JS, class S>
DEV uint32_t pb_array(S& s, uint32_t tag, uint32_t elem, const u8* txt, uint32_t tn) {
  if (elem == ETLG_TC_JSON && !JS) return RB_E_HOST_CELL;
  if (JS && elem == ETLG_TC_JSON) {   // one string field per element: j.to_string() behind reject_nulls and validate_elements(validate_json_for_bigquery) (validation.rs:185-188)
    // Which report a cell with several problems gets: an element too long to look at makes the cell the host's (it may not even be JSON);
    // then the decode error (an element that is not JSON); then the sink's own, in the reference's order — reject_nulls over the whole
    // array, validate_elements after it; an element beyond json_display's limits last (it could only add the integer rule's report).
    u8 tmp[kJsonElemMax];
    uint32_t cnt = 0;
    bool too_long = false, bad_json = false, has_null = false, bq = false, limit = false;
    if (arr_spans(txt, tn, cnt, [&](uint32_t, bool is_null, uint32_t p0, uint32_t p1, uint32_t ulen) {
          if (is_null) { has_null = true; return; }
          if (ulen > kJsonElemMax) { too_long = true; return; }
          uint32_t k = 0;
          arr_unescape(txt, p0, p1, [&](u8 c) { tmp[k++] = c; });
          if (std::is_same<S, RbCount>::value && !json_valid(tmp, ulen)) { bad_json = true; return; }
          JsCount c;
          const uint32_t e = json_display(c, tmp, ulen, true);
          if (e) { if (e == JD_BQ_INT) bq = true; else limit = true; return; }
          if (too_long | bad_json | has_null | bq | limit) return;   // (nothing of this row will be kept)
          pb_key(s, tag, 2); s.varint64(c.n);
          if (std::is_same<S, RbCount>::value) s.zeros(c.n); else (void)json_display(s, tmp, ulen, false);
        })) return RB_E_HOST_CELL;
    return too_long ? RB_E_HOST_CELL : bad_json ? RB_E_JSON : has_null ? RB_E_BQ_ARRAY_NULL : bq ? RB_E_BQ_NUMERIC_SCALE : limit ? RB_E_HOST_CELL : 0u;
  }
  if (elem == ETLG_TC_STRING || elem == ETLG_TC_BYTEA) {   // one string / bytes field per element: the unescaped bytes / the decoded bytes
    uint32_t cnt = 0;
    bool has_null = false, bad = false;   // (a bytea element that is not "\x" + hex pairs is the reference's decode error — the host raises it — and comes before the sink's NULL rule)
    if (arr_spans(txt, tn, cnt, [&](uint32_t, bool is_null, uint32_t p0, uint32_t p1, uint32_t ulen) {
          if (is_null) { has_null = true; return; }
          if (elem == ETLG_TC_STRING) { if (!has_null) { pb_key(s, tag, 2); s.varint64(ulen); arr_unescape(txt, p0, p1, [&](u8 c) { s.put(c); }); } return; }
          const uint32_t nb = arr_bytea_len(txt, p0, p1, ulen);
          if (nb == ~0u) { bad = true; return; }
          if (has_null | bad) return;
          pb_key(s, tag, 2); s.varint64(nb);
          uint32_t k = 0, hi = 0;
          arr_unescape(txt, p0, p1, [&](u8 c) { if (k >= 2) { const uint32_t h = (uint32_t)arr_hexv(c); if (k & 1) s.put((u8)((hi << 4) | h)); else hi = h; } k++; });
        })) return RB_E_HOST_CELL;
    return bad ? RB_E_HOST_CELL : has_null ? RB_E_BQ_ARRAY_NULL : 0u;
  }
  const bool packed = elem == ETLG_TC_BOOL || elem == ETLG_TC_I16 || elem == ETLG_TC_I32 || elem == ETLG_TC_U32 || elem == ETLG_TC_I64 ||
                      elem == ETLG_TC_F32 || elem == ETLG_TC_F64 || elem == ETLG_TC_TIMESTAMPTZ;
  auto none = [](uint32_t) -> u8* { return nullptr; };
  auto value64 = [&](const uint32_t* w) -> uint64_t {   // what the varint of an element holds
    if (elem == ETLG_TC_I16 || elem == ETLG_TC_I32) return (uint64_t)(int64_t)(int32_t)w[0];
    if (elem == ETLG_TC_I64) return ((uint64_t)w[1] << 32) | w[0];
    if (elem == ETLG_TC_TIMESTAMPTZ) return (uint64_t)((((int64_t)(int32_t)w[0] - kCeDays1970) * 86400 + (int64_t)w[1]) * 1000000 + (int64_t)(w[2] / 1000u));
    return (uint64_t)w[0];   // bool, oid
  };
  uint32_t cnt = 0, nulls = 0, plen = 0;
  bool scale_bad = false;
  if (arr_walk<false>(txt, tn, elem, cnt, [&](uint32_t, bool is_null, const uint32_t* w, const u8* scratch) {
        if (is_null) { nulls++; return; }
        if (elem == ETLG_TC_F32) plen += 4; else if (elem == ETLG_TC_F64) plen += 8; else if (elem == ETLG_TC_BOOL) plen += 1;
        else if (packed) { uint64_t v = value64(w); do { plen++; v >>= 7; } while (v); }
        else if (elem == ETLG_TC_NUMERIC) {   // validate_elements(validate_numeric_for_bigquery) behind reject_nulls (validation.rs:169-172)
          const u8* ent = scratch + w[0];
          if (ent[0] == ETLG_NUM_VALUE && ((uint32_t)ent[4] | ((uint32_t)ent[5] << 8)) > 38u) scale_bad = true;
        }
      }, none)) return RB_E_HOST_CELL;
  if (nulls) return RB_E_BQ_ARRAY_NULL;
  if (scale_bad) return RB_E_BQ_NUMERIC_SCALE;
  if (!cnt) return 0;
  if (packed) { pb_key(s, tag, 2); s.varint64(plen); }
  (void)arr_walk<false>(txt, tn, elem, cnt, [&](uint32_t, bool, const uint32_t* w, const u8* scratch) {
    switch (elem) {
      case ETLG_TC_NUMERIC: { const u8* ent = scratch + w[0]; pb_key(s, tag, 2); s.varint64(numeric_str_len(ent)); numeric_str(s, ent); break; }
      case ETLG_TC_TIMETZ: pb_key(s, tag, 2); s.varint64(timetz_str_len((const u8*)w)); timetz_str(s, (const u8*)w); break;
      case ETLG_TC_BOOL: s.put(w[0] ? 1 : 0); break;
      case ETLG_TC_F32: s.put32(w[0]); break;
      case ETLG_TC_F64: s.put64(((uint64_t)w[1] << 32) | w[0]); break;
      case ETLG_TC_DATE: pb_key(s, tag, 2); s.varint64(10); pb_date(s, (int32_t)w[0]); break;
      case ETLG_TC_TIME: pb_key(s, tag, 2); s.varint64(8 + time_frac_len(w[1])); time_str(s, w[0], w[1]); break;
      case ETLG_TC_TIMESTAMP: pb_key(s, tag, 2); s.varint64(19 + time_frac_len(w[2])); pb_date(s, (int32_t)w[0]); s.put(' '); time_str(s, w[1], w[2]); break;
      case ETLG_TC_UUID: {
        const u8* b16 = (const u8*)w;
        pb_key(s, tag, 2); s.varint64(36);
        for (int k = 0; k < 16; k++) {
          const uint32_t b = b16[k], h = b >> 4, l = b & 15;
          if (k == 4 || k == 6 || k == 8 || k == 10) s.put('-');
          s.put((u8)(h < 10 ? '0' + h : 'a' + h - 10)); s.put((u8)(l < 10 ? '0' + l : 'a' + l - 10));
        }
        break;
      }
      default: s.varint64(value64(w)); break;
    }
  }, none);
  return 0;
}

template <bool JS, class S, class M>
DEV uint32_t pb_row(const RbJob& j, uint64_t r, S& s, uint32_t c_lo, uint32_t c_hi, M&& mark) {
  // (the row's kind sits in the top bits of its base: ColSel / pb_selected)
  const unsigned long long rbase = j.row_base[r];
  const bool del = (rbase & kPbDelete) != 0, keyimg = (rbase & kPbKey) != 0;
  const uint64_t base = rbase & kPbBase;
  uint32_t err0 = 0;
  for (uint32_t i = c_lo; i < c_hi; i++) {
    mark(i);
    const uint32_t cd = j.cols[i], cls = cd & 0xFF, tag = i + 1;
    uint32_t off = cd >> 16, sti = i;
    if (del) {  // bigquery_delete_row (core.rs:1742-1754): only the primary-key cells of the old image, under their column tags
      const uint32_t kc = j.kcols[i];
      if (!(kc & 4u)) continue;
      if (keyimg) { off = kc >> 16; sti = (kc >> 8) & 0xFFu; }
    }
    const uint32_t st = (j.fixed[base + sti / 4] >> (2 * (sti % 4))) & 3u;
    if (st == ETLG_CELL_NULL) continue;                       // Cell::Null => {}
    if (st != ETLG_CELL_VALUE && !((cls == ETLG_TC_JSON || cls == ETLG_TC_ARRAY) && st == ETLG_CELL_DEFERRED)) { if (!err0) err0 = (i << 8) | RB_E_HOST_CELL; continue; }
    const u8* slot = j.fixed + base + off;
    const uint32_t w0 = ld32a(slot);
    if (cls == ETLG_TC_ARRAY) {
      if (const uint32_t e = pb_array<JS>(s, tag, (cd >> 9) & 0x7Fu, j.heap + w0, ld32a(slot + 4))) {
        if (e == RB_E_JSON) return (i << 8) | e;
        if (!err0) err0 = (i << 8) | e;
      }
      continue;
    }
    switch (cls) {
      case ETLG_TC_BOOL: pb_key(s, tag, 0); s.put(w0 ? 1 : 0); break;
      case ETLG_TC_I16: case ETLG_TC_I32: pb_key(s, tag, 0); s.varint64((uint64_t)(int64_t)(int32_t)w0); break;
      case ETLG_TC_I64: pb_key(s, tag, 0); s.varint64(((uint64_t)ld32a(slot + 4) << 32) | w0); break;
      case ETLG_TC_U32: pb_key(s, tag, 0); s.varint64((uint64_t)w0); break;
      case ETLG_TC_F32: pb_key(s, tag, 5); s.put32(w0); break;
      case ETLG_TC_F64: pb_key(s, tag, 1); s.put64(((uint64_t)ld32a(slot + 4) << 32) | w0); break;
      case ETLG_TC_STRING: case ETLG_TC_BYTEA: { const uint32_t len = ld32a(slot + 4); pb_key(s, tag, 2); s.varint64(len); s.bytes(j.heap + w0, len); break; }
      case ETLG_TC_DATE: pb_key(s, tag, 2); s.varint64(10); pb_date(s, (int32_t)w0); break;
      case ETLG_TC_TIME: { const uint32_t ns = ld32a(slot + 4); pb_key(s, tag, 2); s.varint64(8 + time_frac_len(ns)); time_str(s, w0, ns); break; }
      case ETLG_TC_TIMESTAMP: {  // "%Y-%m-%d %H:%M:%S%.f" (TIMESTAMP_FORMAT :21)
        const uint32_t secs = ld32a(slot + 4), ns = ld32a(slot + 8);
        pb_key(s, tag, 2); s.varint64(19 + time_frac_len(ns)); pb_date(s, (int32_t)w0); s.put(' '); time_str(s, secs, ns); break;
      }
      case ETLG_TC_TIMESTAMPTZ: {  // epoch microseconds as int64 (:176-179)
        const int64_t days = (int64_t)(int32_t)w0 - kCeDays1970;
        pb_key(s, tag, 0); s.varint64((uint64_t)((days * 86400 + (int64_t)ld32a(slot + 4)) * 1000000 + (int64_t)(ld32a(slot + 8) / 1000u))); break;
      }
      case ETLG_TC_UUID:  // Uuid Display: hyphenated lowercase
        pb_key(s, tag, 2); s.varint64(36);
        for (int k = 0; k < 16; k++) {
          const uint32_t b = slot[k], h = b >> 4, l = b & 15;
          if (k == 4 || k == 6 || k == 8 || k == 10) s.put('-');
          s.put((u8)(h < 10 ? '0' + h : 'a' + h - 10)); s.put((u8)(l < 10 ? '0' + l : 'a' + l - 10));
        }
        break;
      case ETLG_TC_TIMETZ: pb_key(s, tag, 2); s.varint64(timetz_str_len(slot)); timetz_str(s, slot); break;   // t.to_string() (:158-161)
      case ETLG_TC_NUMERIC: {  // n.to_string() (:146-149) behind validate_numeric_for_bigquery (bigquery/validation.rs:20-35): more than 38 decimal places would be rounded
        const u8* ent = j.heap + w0;
        if (ent[0] == ETLG_NUM_VALUE && ((uint32_t)ent[4] | ((uint32_t)ent[5] << 8)) > 38u) { if (!err0) err0 = (i << 8) | RB_E_BQ_NUMERIC_SCALE; break; }
        pb_key(s, tag, 2); s.varint64(numeric_str_len(ent)); numeric_str(s, ent); break;
      }
      case ETLG_TC_JSON: {  // j.to_string() (:173-176) behind validate_json_for_bigquery (bigquery/validation.rs:47-85)
        if (!JS) { if (!err0) err0 = (i << 8) | RB_E_HOST_CELL; break; }
        if (const uint32_t e = rb_json(s, j.heap + w0, ld32a(slot + 4), true, [&](uint32_t len) { pb_key(s, tag, 2); s.varint64(len); })) {
          if (e == RB_E_JSON) return (i << 8) | e;   // (the reference's decode fails before the sink validates anything: it beats an earlier cell's report)
          if (!err0) err0 = (i << 8) | e;
        }
        break;
      }
      default: if (!err0) err0 = (i << 8) | RB_E_HOST_CELL; break;   // arrays: packed / repeated fields, host-side validation
    }
  }
  if (err0) return err0;
  if (c_hi != j.n_cols) return 0;
  const uint64_t ev = j.row_event[r];
  pb_key(s, j.n_cols + 1, 2); s.varint64(6);
  { const char* op = del ? "DELETE" : "UPSERT"; for (int k = 0; k < 6; k++) s.put((u8)op[k]); }
  pb_key(s, j.n_cols + 2, 2); s.varint64(50);
  pb_hex16(s, j.ev_commit[ev]); s.put('/'); pb_hex16(s, j.ev_ord[ev]); s.put('/'); pb_hex16(s, (rbase & kPbSecond) ? 1 : 0);   // bigquery_sequence_key (:1405-1407)
  return 0;
}

// JS: the table has a json column (kernels of their own, as for the Arrow columns)
template <bool JS>
__global__ __launch_bounds__(256) void k_rb_lens(RbJob j, unsigned long long* blk) {
  __shared__ uint64_t lds_sum[4];
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  RbCount c;
  if (r < j.n_rows) {
    // (where pieces 1 .. qparts-1 of the row begin, for the byte pass: piece q starts at column q * n_cols / qparts)
    uint32_t next = 1;
    auto mark = [&](uint32_t i) {
      while (next < j.qparts && i == rb_part_col(j, next)) { j.part_off[(uint64_t)(next - 1) * j.n_rows + r] = c.n; next++; }
    };
    const uint32_t e = j.format ? pb_row<JS>(j, r, c, 0, j.n_cols, mark) : rb_row<JS>(j, r, c, 0, j.n_cols, mark);
    // first failing row in event order, rows with a date out of range before all others (bit 62 clear)
    // (and a json cell that is not JSON before those: the reference's decode fails before any sink sees a row)
    if (e) {
      const unsigned long long rank = (e & 0xFFu) == RB_E_JSON ? 0ull : ((e & 0xFFu) == RB_E_DATE_RANGE || j.format) ? 1ull << 61 : 1ull << 62;
      atomicMin(j.err, rank | (unsigned long long)((r << 24) | e)); c.n = 0;
    }
    j.lens[r] = c.n;
  }
  const uint64_t t = block_sum64(c.n, lds_sum);   // (the block's sum for the offsets scan)
  if (threadIdx.x == 0) blk[blockIdx.x] = t;
}

// The byte pass. One lane per row is few waves for what each has to do — a 64 MiB cfg3 batch is 175 000 rows of 270 bytes: 2.7 waves per
// SIMD, each a serial chain of loads and stores (132 us; profiles/r05v_rb_rows_ablation.txt) — so a row is split among `parts` lanes
// (1-4, the host picks it from the row count), each writing the columns [part * n / parts, (part + 1) * n / parts) from the byte offset
// the counting pass noted. A wave holds 64 (128, 256) consecutive rows of ONE part — the lanes walk the same columns — and the parts of
// a row sit in ONE workgroup: with a part per workgroup (blockIdx.y) every cache line of the output was written from several XCDs, and
// the kernel got slower, not faster (186 us against 134; profiles/r05y_rb_rows_parts.txt).
DEV uint32_t rb_rows_per_block(uint32_t parts) { return parts == 1 ? 256u : parts == 2 ? 128u : 64u; }
// ... and the lanes do not store to global memory themselves: 64 lanes x 8 bytes at a stride of a row is 64 write requests per
// instruction (the request rate, not the bytes, bounded the kernel). The workgroup's rows are one contiguous piece of the output: when it
// fits kRbLds, the lanes build it in LDS and the whole workgroup stores it in 16-byte pieces; a piece that does not fit (rows of more
// than ~500 bytes on average) is written directly as before.
constexpr uint32_t kRbLds = 32 * 1024;
template <bool JS>
__global__ __launch_bounds__(256) void k_rb_rows(RbJob j) {
  __shared__ uint4 img[kRbLds / 16 + 2];
  const uint32_t rpb = rb_rows_per_block(j.parts), part = threadIdx.x / rpb;
  const uint64_t r0 = (uint64_t)blockIdx.x * rpb, r = r0 + threadIdx.x % rpb;
  const uint64_t r1 = r0 + rpb < j.n_rows ? r0 + rpb : j.n_rows;
  const uint64_t g0 = (uint64_t)j.offsets[r0], g1 = (uint64_t)j.offsets[r1];
  const uint32_t pad = (uint32_t)((uintptr_t)(j.out + g0) & 15u);   // the image starts at the 16-byte line the piece starts in
  const bool staged = g1 - g0 + pad <= kRbLds;                      // (uniform in the workgroup)
  const bool active = part < j.parts && r < j.n_rows && j.lens[r];
  const uint32_t q = part * (j.qparts / j.parts);   // the lane's first piece
  const uint32_t c_lo = rb_part_col(j, q), c_hi = part + 1 >= j.parts ? j.n_cols : rb_part_col(j, q + j.qparts / j.parts);
  const uint32_t po = active && q ? j.part_off[(uint64_t)(q - 1) * j.n_rows + r] : 0u;
  auto none = [](uint32_t) {};
  if (!staged) {
    if (active) {
      RbWrite w(j.out + j.offsets[r] + po);
      if (j.format) (void)pb_row<JS>(j, r, w, c_lo, c_hi, none); else (void)rb_row<JS>(j, r, w, c_lo, c_hi, none);
      w.finish();
    }
    return;
  }
  const uint32_t total = pad + (uint32_t)(g1 - g0), nch = (total + 15u) / 16u;
  for (uint32_t k = threadIdx.x; k < nch; k += 256) img[k] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (active) {
    const uint32_t o = pad + (uint32_t)((uint64_t)j.offsets[r] - g0) + po;
    RbLdsWrite w((uint32_t*)img + (o >> 2), o & 3u);
    if (j.format) (void)pb_row<JS>(j, r, w, c_lo, c_hi, none); else (void)rb_row<JS>(j, r, w, c_lo, c_hi, none);
    w.finish();
  }
  __syncthreads();
  u8* gb = j.out + g0 - pad;
  for (uint32_t k = threadIdx.x; k < nch; k += 256) {
    const uint32_t b0 = k * 16u;
    if (b0 >= pad && b0 + 16u <= total) *(uint4*)(gb + b0) = img[k];
    else for (uint32_t b = b0 < pad ? pad : b0; b < b0 + 16u && b < total; b++) gb[b] = ((const u8*)img)[b];   // the first / last line: the bytes outside belong to the neighbours
  }
}


// ---- Event::size_hint (crates/etl/src/event.rs:295-320; data/table_row.rs:248-299)
DEV uint64_t hint_row(const HintJob& j, const uint32_t* sl, uint64_t base, bool key, bool& incomplete) {
  const uint32_t n_cols = sl[0], n_ident = sl[1], cb = sl[4];
  uint64_t total = (uint64_t)j.m_row + (uint64_t)(key ? n_ident : n_cols) * j.m_cell;   // TableRow + Vec<Cell> capacity
  uint32_t k = 0;
  for (uint32_t i = 0; i < n_cols; i++) {
    const uint32_t cd = j.cols[2 * (cb + i)], cls = cd & 0xFF;
    if (key && !((cd >> 8) & 1)) continue;
    const uint32_t pos = key ? k : i, off = key ? j.cols[2 * (cb + i) + 1] : cd >> 16;
    k++;
    const uint32_t st = (j.fixed[base + pos / 4] >> (2 * (pos % 4))) & 3u;
    if (st == ETLG_CELL_NULL) continue;
    if (st == ETLG_CELL_MISSING) { incomplete = true; continue; }
    const bool text_form = cls == ETLG_TC_JSON || cls == ETLG_TC_ARRAY || cls == ETLG_TC_NUMERIC;
    if (st == ETLG_CELL_DEFERRED) { if (text_form) incomplete = true; continue; }   // a deferred fixed-width cell owns no heap
    const u8* slot = j.fixed + base + off;
    if (cls == ETLG_TC_STRING || cls == ETLG_TC_BYTEA) total += ld32a(slot + 4);
    else if (cls == ETLG_TC_NUMERIC) { const u8* h = j.heap + ld32a(slot); if (h[0] == ETLG_NUM_VALUE) total += 2u * (uint32_t)(h[6] | (h[7] << 8)); }
    else if (cls == ETLG_TC_JSON || cls == ETLG_TC_ARRAY) incomplete = true;
  }
  return total;
}

__global__ __launch_bounds__(256) void k_size_hints(HintJob j) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= j.n_events) return;
  const uint32_t kind = j.ev_kind[i], fl = j.ev_flags[i];
  uint64_t v = 0;
  bool inc = false;
  if (kind == 'B') v = j.m_begin;
  else if (kind == 'C') v = j.m_commit;
  else if (kind == 'R') v = j.m_relation;
  else if (kind == 'T') v = (uint64_t)j.m_truncate + (uint64_t)j.ev_table[i] * j.m_rts;
  else if (kind == 'I' || kind == 'U' || kind == 'D') {
    const uint32_t s = j.ev_slot[i];
    if (s >= j.n_slots) { inc = true; }
    else {
      const uint32_t* sl = j.slots + 5 * s;
      uint64_t base = j.ev_body[i];
      v = kind == 'I' ? j.m_insert : kind == 'U' ? j.m_update : j.m_delete;
      if (kind != 'I') {
        const uint32_t ok = fl & 3u;
        if (ok) { v += hint_row(j, sl, base, ok == ETLG_OLD_KEY, inc); base += ok == ETLG_OLD_KEY ? sl[3] : sl[2]; }
      }
      if (kind != 'D') {
        if (fl & ETLG_FLAG_PARTIAL) inc = true;
        v += hint_row(j, sl, base, false, inc);
      }
    }
  }
  j.out[i] = v | (inc ? (1ull << 63) : 0ull);
}

// ---- the finish pass (etlg_batch_finish_cells, include/etlg.h): cells a decode left ETLG_CELL_DEFERRED are settled in the arena itself.
// Array literals (parse_cell_from_postgres_text_array, crates/etl/src/postgres/codec/text.rs:163-312) become typed heap entries — header,
// validity bits, element slots or end offsets + bytes (include/etlg.h: etlg_array_hdr) — appended behind the batch's heap; float texts the
// fast rule could not decide get their exactly rounded bits (float_slow.h). One thread per (event, row image, finishable column); a cell
// the device does not settle (a malformed literal, a json element, a numeric element of more than 40 characters) stays DEFERRED for the
// host, exactly as before. Two walks of the text like k_arr_count / k_arr_fill: sizes -> exclusive scan -> entries.
enum : uint32_t { FIN_ARRAYS = 1u, FIN_FLOATS = 2u };

struct FinCell { bool on; uint32_t cls, elem; u8* slot; uint32_t* stw; uint32_t stsh; };
DEV FinCell fin_cell(const FinJob& j, uint64_t t) {
  FinCell c{false, 0, 0, nullptr, nullptr, 0};
  const uint32_t K = 2u * j.maxfin;
  const uint64_t ev = t / K;
  const uint32_t r = (uint32_t)(t % K), img = r / j.maxfin, q = r % j.maxfin;
  if (ev >= j.n_events) return c;
  const uint32_t kind = j.ev_kind[ev];
  if (!(kind == 'I' || kind == 'U' || kind == 'D')) return c;
  const uint32_t s = j.ev_slot[ev];
  if (s >= j.n_slots) return c;
  const uint32_t* sl = j.slots + 7 * s;
  if (q >= sl[6]) return c;
  const uint32_t col = j.fin[sl[5] + q];
  const uint32_t* cd = j.cols + 3 * (size_t)(sl[4] + col);
  const uint32_t ok = kind == 'I' ? 0u : (uint32_t)j.ev_flags[ev] & 3u;
  uint64_t base = j.ev_body[ev];
  uint32_t pos = col, off = cd[1] & 0xFFFFu;
  if (img == 0) {
    if (ok == ETLG_OLD_NONE) return c;
    if (ok == ETLG_OLD_KEY) { if (!((cd[0] >> 8) & 1u)) return c; pos = cd[2]; off = cd[1] >> 16; }
  } else {
    if (kind == 'D') return c;
    base += ok == ETLG_OLD_KEY ? sl[3] : ok == ETLG_OLD_FULL ? sl[2] : 0u;
  }
  u8* stb = j.fixed + base + pos / 4;
  const uint32_t st = (*stb >> (2 * (pos % 4))) & 3u;
  if (st != ETLG_CELL_DEFERRED) return c;
  c.on = true; c.cls = cd[0] & 0xFFu; c.elem = (cd[0] >> 16) & 0xFFu;
  c.slot = j.fixed + base + off;
  c.stw = (uint32_t*)((uintptr_t)stb & ~(uintptr_t)3);
  c.stsh = 8u * (uint32_t)((uintptr_t)stb & 3u) + 2u * (pos % 4);
  return c;
}
DEV bool fin_elem_fixed(uint32_t e) { return e == ETLG_TC_BOOL || e == ETLG_TC_I16 || e == ETLG_TC_I32 || e == ETLG_TC_I64 || e == ETLG_TC_U32 || e == ETLG_TC_F32 || e == ETLG_TC_F64 ||
                                             e == ETLG_TC_DATE || e == ETLG_TC_TIME || e == ETLG_TC_TIMETZ || e == ETLG_TC_TIMESTAMP || e == ETLG_TC_TIMESTAMPTZ || e == ETLG_TC_UUID; }
DEV bool fin_elem_var(uint32_t e) { return e == ETLG_TC_STRING || e == ETLG_TC_BYTEA || e == ETLG_TC_NUMERIC; }

// bytes of the typed entry of one array literal, 0 = not settled here
DEV uint32_t fin_array_bytes(const u8* s, uint32_t n, uint32_t elem, bool exact, uint32_t& cnt) {
  cnt = 0;
  uint64_t data = 0;
  auto none = [](uint32_t) -> u8* { return nullptr; };
  uint32_t e;
  if (fin_elem_fixed(elem)) e = arr_walk<false>(s, n, elem, cnt, [](uint32_t, bool, const uint32_t*, const u8*) {}, none, exact);
  else if (elem == ETLG_TC_NUMERIC) e = arr_walk<false>(s, n, elem, cnt, [&](uint32_t, bool is_null, const uint32_t* w, const u8*) { if (!is_null) data += pad4(w[1]); }, none);
  else if (fin_elem_var(elem)) e = arr_walk<true>(s, n, elem, cnt, [&](uint32_t, bool, const uint32_t* w, const u8*) { data += w[0]; }, none);
  else return 0;
  if (e) return 0;
  const uint64_t tot = 8ull + 4ull * ((cnt + 31u) >> 5) + (fin_elem_fixed(elem) ? (uint64_t)cnt * slot_bytes(elem) : 4ull * cnt + ((data + 3ull) & ~3ull));
  return tot > 0x7FFFFFF0ull ? 0u : (uint32_t)tot;
}

// Thread u of the grid takes cell t = event * K + row, with u = row * n_events + event: the lanes of a wave hold the SAME column of
// consecutive events (one element class, one code path: with t = u every lane of a wave decoded another class, 31 paths one after the
// other — 7.2 ms for the fill of a 30 MB type-matrix batch), while sizes and entries stay in (event, image, column) order.
DEV uint64_t fin_thread_cell(const FinJob& j, uint64_t u) {
  const uint64_t K = 2ull * j.maxfin;
  return (u % j.n_events) * K + u / j.n_events;
}

__global__ __launch_bounds__(256) void k_fin_count(FinJob j) {
  const uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (u >= j.n_events * 2ull * j.maxfin) return;
  const uint64_t t = fin_thread_cell(j, u);
  const FinCell c = fin_cell(j, t);
  uint32_t len = 0;
  uint32_t cnt = 0;
  if (c.on && c.cls == ETLG_TC_ARRAY && (j.what & FIN_ARRAYS)) len = fin_array_bytes(j.heap + ld32a(c.slot), ld32a(c.slot + 4), c.elem, (j.what & FIN_FLOATS) != 0, cnt);
  j.lens[t] = len;
  if (len) j.counts[t] = cnt;
}

DEV void fin_tally(const FinJob& j, bool seen, bool arr, bool flt, bool left) {   // one atomic per wave and counter
  const unsigned long long ms = __ballot(seen), ma = __ballot(arr), mf = __ballot(flt), ml = __ballot(left);
  if ((threadIdx.x & 63u) == 0) {
    if (ms) atomicAdd(&j.stats[0], (unsigned long long)__builtin_popcountll(ms));
    if (ma) atomicAdd(&j.stats[1], (unsigned long long)__builtin_popcountll(ma));
    if (mf) atomicAdd(&j.stats[2], (unsigned long long)__builtin_popcountll(mf));
    if (ml) atomicAdd(&j.stats[3], (unsigned long long)__builtin_popcountll(ml));
  }
}

__global__ __launch_bounds__(256) void k_fin_fill(FinJob j) {
  const uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const bool in_grid = u < j.n_events * 2ull * j.maxfin;
  const uint64_t t = in_grid ? fin_thread_cell(j, u) : 0;
  FinCell c{false, 0, 0, nullptr, nullptr, 0};
  if (in_grid) c = fin_cell(j, t);
  bool settled = false, was_arr = false, was_flt = false;
  if (c.on) {
  const u8* s = j.heap + ld32a(c.slot);
  const uint32_t n = ld32a(c.slot + 4);
  if ((c.cls == ETLG_TC_F32 || c.cls == ETLG_TC_F64) && (j.what & FIN_FLOATS)) {
    // (the decode validated the grammar: only texts parse_float_fast calls inconclusive are DEFERRED)
    const uint64_t bits = parse_float_exact_t([&](uint32_t i) { return (uint32_t)s[i]; }, n, c.cls == ETLG_TC_F32);
    ((uint32_t*)c.slot)[0] = (uint32_t)bits; ((uint32_t*)c.slot)[1] = (uint32_t)(bits >> 32);
    settled = true; was_flt = true;
  } else if (c.cls == ETLG_TC_ARRAY && (j.what & FIN_ARRAYS) && j.lens && j.lens[t]) {
    const uint32_t bytes = j.lens[t], elem = c.elem;
    const uint64_t at = j.heap_base + (uint64_t)j.offsets[t];
    uint32_t* const e32 = (uint32_t*)(j.heap + at);
    uint32_t cnt = j.counts[t];   // (from the count pass: the header and the validity words come before the elements)
    auto none = [](uint32_t) -> u8* { return nullptr; };
    const bool fixed = fin_elem_fixed(elem), exact = (j.what & FIN_FLOATS) != 0;
    const uint32_t total = cnt, vw = (total + 31u) >> 5, sb = fixed ? slot_bytes(elem) : 0u;
    e32[0] = total; e32[1] = elem | (sb << 8);
    for (uint32_t i = 0; i < vw; i++) e32[2 + i] = 0;
    uint32_t* const valid = e32 + 2;
    if (fixed) {
      uint32_t* const vals = e32 + 2 + vw;
      (void)arr_walk<false>(s, n, elem, cnt, [&](uint32_t k, bool is_null, const uint32_t* w, const u8*) {
        uint32_t* d = vals + (size_t)k * (sb >> 2);
        for (uint32_t q = 0; q < (sb >> 2); q++) d[q] = is_null ? 0u : w[q];
        if (!is_null) valid[k >> 5] |= 1u << (k & 31u);
      }, none, exact);
    } else {
      uint32_t* const ends = e32 + 2 + vw;
      u8* const data = (u8*)(ends + total);
      uint32_t run = 0;
      if (elem == ETLG_TC_NUMERIC) {
        (void)arr_walk<false>(s, n, elem, cnt, [&](uint32_t k, bool is_null, const uint32_t* w, const u8* scratch) {
          if (!is_null) {
            const uint32_t nb = pad4(w[1]);
            heap_copy(data + run, scratch + w[0], w[1]);
            run += nb;
            valid[k >> 5] |= 1u << (k & 31u);
          }
          ends[k] = run;
        }, none);
      } else {
        // text / bytea elements: their bytes leave as the walk unescapes them; element k starts where element k - 1 ended
        uint32_t lens_run = 0;
        (void)arr_walk<true>(s, n, elem, cnt, [&](uint32_t k, bool is_null, const uint32_t* w, const u8*) {
          lens_run += w[0];
          ends[k] = lens_run;
          if (!is_null) valid[k >> 5] |= 1u << (k & 31u);
        }, [&](uint32_t) -> u8* { return data + lens_run; });
        run = lens_run;
        while (run & 3u) data[run++] = 0;
      }
    }
    ((uint32_t*)c.slot)[0] = (uint32_t)at; ((uint32_t*)c.slot)[1] = bytes;
    settled = true; was_arr = true;
  }
  if (settled) atomicAnd(c.stw, ~(3u << c.stsh));   // DEFERRED (3) -> VALUE (0); the other cells of the row share the word
  }
  fin_tally(j, c.on, was_arr, was_flt, c.on && !settled);
}

}  // namespace etlg

extern "C" {

using namespace etlg;

void etlg_k_col_select(const void* selv, hipStream_t st) {
  const ColSel s = *(const ColSel*)selv;
  if (!s.nblocks) return;
  hipLaunchKernelGGL(k_col_count, dim3(s.nblocks), dim3(256), 0, st, s);
  hipLaunchKernelGGL(k_col_scan, dim3(1), dim3(256), 0, st, s.blk, s.nblocks);
  hipLaunchKernelGGL(k_col_rows, dim3(s.nblocks), dim3(256), 0, st, s);
}

void etlg_k_col_fixed(const void* jv, hipStream_t st) {
  const ColJob j = *(const ColJob*)jv;
  if (j.n_rows) hipLaunchKernelGGL(k_col_fixed, dim3((uint32_t)((j.n_rows + 255) / 256)), dim3(256), 0, st, j);
}

// Packed forms (jobs: n <= etlg_k_col_pack_max() ColJob records of ONE hand-off, all with the same n_rows): every fixed-width column /
// pass 1 of every var-len column (lens -> block sums -> offsets; blk[i] / offs[i]: the column's scan scratch and offsets) / pass 2.
uint32_t etlg_k_col_pack_max(void) { return kPack; }
static void fill_pack(ColPack& p, const ColJob* jobs, uint32_t n, unsigned long long* const* blk, int64_t* const* offs, unsigned long long* const* tot = nullptr) {
  for (uint32_t i = 0; i < n; i++) { p.j[i] = jobs[i]; p.blk[i] = blk ? blk[i] : nullptr; p.offs[i] = offs ? offs[i] : nullptr; p.tot[i] = tot ? tot[i] : nullptr; }
}
void etlg_k_col_fixed_pack(const void* jobs, uint32_t n, hipStream_t st) {
  const ColJob* j = (const ColJob*)jobs;
  if (!n || !j[0].n_rows) return;
  ColPack p; fill_pack(p, j, n, nullptr, nullptr);
  hipLaunchKernelGGL(k_col_fixed_pack, dim3((uint32_t)((j[0].n_rows + 255) / 256), n), dim3(256), 0, st, p);
}
void etlg_k_col_var_pack(const void* jobs, uint32_t n, unsigned long long* const* blk, int64_t* const* offs, unsigned long long* const* tot, int step, hipStream_t st) {
  const ColJob* j = (const ColJob*)jobs;
  if (!n || !j[0].n_rows) return;
  const uint32_t nb = (uint32_t)((j[0].n_rows + 255) / 256);
  ColPack p; fill_pack(p, j, n, blk, offs, tot);
  bool js = false;
  for (uint32_t i = 0; i < n; i++) js |= j[i].kind == AK_JSON_STR;
  if (step == 0) {
    if (js) hipLaunchKernelGGL(k_col_lens_pack<true>, dim3(nb, n), dim3(256), 0, st, p); else hipLaunchKernelGGL(k_col_lens_pack<false>, dim3(nb, n), dim3(256), 0, st, p);
    hipLaunchKernelGGL(k_col_len_scan_pack, dim3(n), dim3(256), 0, st, p, nb);
    hipLaunchKernelGGL(k_col_offsets_pack, dim3(nb, n), dim3(256), 0, st, p);
  } else {
    if (js) hipLaunchKernelGGL(k_col_var2_pack<true>, dim3(nb, n), dim3(256), 0, st, p); else hipLaunchKernelGGL(k_col_var2_pack<false>, dim3(nb, n), dim3(256), 0, st, p);
  }
}

// blk: (nblocks + 1) x u64 scratch
void etlg_k_col_var(const void* jv, unsigned long long* blk, int64_t* offsets, int step, hipStream_t st) {
  const ColJob j = *(const ColJob*)jv;
  if (!j.n_rows) return;
  const uint32_t nb = (uint32_t)((j.n_rows + 255) / 256);
  if (step == 0) {
    if (j.kind == AK_JSON_STR) hipLaunchKernelGGL(k_col_lens<true>, dim3(nb), dim3(256), 0, st, j, blk); else hipLaunchKernelGGL(k_col_lens<false>, dim3(nb), dim3(256), 0, st, j, blk);
    hipLaunchKernelGGL(k_col_len_scan, dim3(1), dim3(256), 0, st, blk, nb);
    hipLaunchKernelGGL(k_col_offsets, dim3(nb), dim3(256), 0, st, (const uint32_t*)j.lens, j.n_rows, (const unsigned long long*)blk, offsets, (unsigned long long*)nullptr);
  } else {
    if (j.kind == AK_JSON_STR) hipLaunchKernelGGL(k_col_fmt<true>, dim3(nb), dim3(256), 0, st, j);
    else if (j.kind == AK_NUMERIC_STR || j.kind == AK_TIMETZ_STR) hipLaunchKernelGGL(k_col_fmt<false>, dim3(nb), dim3(256), 0, st, j);
    else hipLaunchKernelGGL(k_col_copy, dim3((uint32_t)((j.n_rows + 255) / 256)), dim3(256), 0, st, j);
  }
}

// lens (u32, n entries) -> offsets (i64, n + 1 entries); blk: (ceil(n / 256) + 1) x u64 scratch
void etlg_k_scan_lens(const uint32_t* lens, uint64_t n, unsigned long long* blk, int64_t* offsets, hipStream_t st) {
  if (!n) return;
  const uint32_t nb = (uint32_t)((n + 255) / 256);
  hipLaunchKernelGGL(k_col_len_blocks, dim3(nb), dim3(256), 0, st, lens, n, blk);
  hipLaunchKernelGGL(k_col_len_scan, dim3(1), dim3(256), 0, st, blk, nb);
  hipLaunchKernelGGL(k_col_offsets, dim3(nb), dim3(256), 0, st, lens, n, (const unsigned long long*)blk, offsets, (unsigned long long*)nullptr);
}

// list columns: step 0 = element counts + list offsets, step 1 = child values / validity
void etlg_k_col_list(const void* jv, unsigned long long* blk, int64_t* offsets, int step, hipStream_t st) {
  const ColJob j = *(const ColJob*)jv;
  if (!j.n_rows) return;
  const uint32_t nb = (uint32_t)((j.n_rows + 255) / 256);
  if (step == 0) {
    hipLaunchKernelGGL(k_arr_count, dim3(nb), dim3(256), 0, st, j);
    hipLaunchKernelGGL(k_col_len_blocks, dim3(nb), dim3(256), 0, st, (const uint32_t*)j.lens, j.n_rows, blk);
    hipLaunchKernelGGL(k_col_len_scan, dim3(1), dim3(256), 0, st, blk, nb);
    hipLaunchKernelGGL(k_col_offsets, dim3(nb), dim3(256), 0, st, (const uint32_t*)j.lens, j.n_rows, (const unsigned long long*)blk, offsets, (unsigned long long*)nullptr);
  } else {
    hipLaunchKernelGGL(k_arr_fill, dim3(nb), dim3(256), 0, st, j);
  }
}

// step 0: lengths + offsets (blk: (nblocks + 1) x u64 scratch); step 1: the bytes
void etlg_k_rowbinary(const void* jv, unsigned long long* blk, int64_t* offsets, unsigned long long* tot, int step, hipStream_t st) {
  const RbJob j = *(const RbJob*)jv;
  if (!j.n_rows) return;
  const uint32_t nb = (uint32_t)((j.n_rows + 255) / 256);
  if (step == 0) {
    if (j.has_json) hipLaunchKernelGGL(k_rb_lens<true>, dim3(nb), dim3(256), 0, st, j, blk); else hipLaunchKernelGGL(k_rb_lens<false>, dim3(nb), dim3(256), 0, st, j, blk);
    hipLaunchKernelGGL(k_col_len_scan, dim3(1), dim3(256), 0, st, blk, nb);
    hipLaunchKernelGGL(k_col_offsets, dim3(nb), dim3(256), 0, st, (const uint32_t*)j.lens, j.n_rows, (const unsigned long long*)blk, offsets, tot);
  } else {
    const uint32_t rpb = j.parts == 1 ? 256u : j.parts == 2 ? 128u : 64u, nbw = (uint32_t)((j.n_rows + rpb - 1) / rpb);
    if (j.has_json) hipLaunchKernelGGL(k_rb_rows<true>, dim3(nbw), dim3(256), 0, st, j); else hipLaunchKernelGGL(k_rb_rows<false>, dim3(nbw), dim3(256), 0, st, j);
  }
}

void etlg_k_size_hints(const void* jv, hipStream_t st) {
  const HintJob j = *(const HintJob*)jv;
  if (j.n_events) hipLaunchKernelGGL(k_size_hints, dim3((uint32_t)((j.n_events + 255) / 256)), dim3(256), 0, st, j);
}

// the finish pass: step 0 = entry sizes of every (event, image, finishable column) + their exclusive scan (blk: (ceil(n / 256) + 1) x u64
// scratch, offsets: n + 1 x i64), step 1 = the entries, the slots and the cell states
void etlg_k_finish(const void* jv, unsigned long long* blk, int64_t* offsets, int step, hipStream_t st) {
  const FinJob j = *(const FinJob*)jv;
  const uint64_t n = j.n_events * 2ull * j.maxfin;
  if (!n) return;
  const uint32_t nb = (uint32_t)((n + 255) / 256);
  if (step == 0) {
    hipLaunchKernelGGL(k_fin_count, dim3(nb), dim3(256), 0, st, j);
    etlg_k_scan_lens(j.lens, n, blk, offsets, st);
  } else hipLaunchKernelGGL(k_fin_fill, dim3(nb), dim3(256), 0, st, j);
}
uint32_t etlg_k_finish_job_bytes(void) { return (uint32_t)sizeof(FinJob); }

}  // extern "C"
