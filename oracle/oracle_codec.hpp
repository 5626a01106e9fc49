// TEST INFRASTRUCTURE — CPU restatement of the reference's text value codec.
//
// This file is part of the parity oracle. It is NOT product code: only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build,
// link or call anything under oracle/.
//
// Every function cites the reference file:line it restates (paths relative to
// the supabase/etl checkout). Library semantics that live in un-vendored
// crates (Rust core int/float/UTF-8 parsing, chrono 0.4.44, uuid 1.23.1,
// serde_json 1.0.149) are restated from their published behaviour; where the
// reference's own known-answer tests do not pin a shape this is said inline.
#pragma once

#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <optional>
#include <string>
#include <string_view>
#include <vector>

#include "../include/etlg.h"

namespace orc {

using u8 = uint8_t;
using sv = std::string_view;

// ---------------------------------------------------------------- value model
// Mirrors `Cell` (crates/etl/src/data/cell.rs:19-57) and `ArrayCell` (:98-134).
// Kept lean on purpose (24-byte payload, one heap block per owned value) so
// that the timed CPU baseline pays roughly what the Rust enum pays: one
// allocation per String/Bytes/Numeric cell and one Vec<Cell> per row.

enum class Tag : u8 {
  Null, Bool, String, I16, I32, U32, I64, F32, F64, Numeric, Date, Time, TimeTz,
  Timestamp, TimestampTz, Uuid, Json, Bytes, Array
};

struct Numeric {  // PgNumeric, crates/etl-postgres/src/numeric.rs:75-96
  u8 kind = ETLG_NUM_VALUE;
  u8 sign = 0;
  int16_t weight = 0;
  uint16_t scale = 0;
  std::vector<int16_t> digits;
};

struct NumBlock {  // one malloc: header + digits (Rust: Vec<i16> inside the enum)
  u8 kind, sign;
  int16_t weight;
  uint16_t scale;
  uint16_t _pad;
  uint32_t ndigits;
  int16_t digits[1];
};

struct Cell;
struct Arr {
  int32_t elem_class = ETLG_TC_STRING;
  std::vector<Cell> elems;  // Tag::Null == None
};

struct Cell {
  Tag tag = Tag::Null;
  union U {
    bool b;
    int64_t i;       // I16/I32/I64/U32
    uint64_t fbits;  // F32 (low 32 bits) / F64 raw IEEE bits
    struct { int32_t date; uint32_t secs, nanos; int32_t offset; } t;
    u8 uuid[16];
    struct { char* p; size_t len; } s;  // String / Bytes / Json(raw text)
    NumBlock* num;
    Arr* arr;
  } u;

  Cell() { memset(&u, 0, sizeof u); }
  Cell(const Cell&) = delete;
  Cell& operator=(const Cell&) = delete;
  Cell(Cell&& o) noexcept : tag(o.tag), u(o.u) { o.tag = Tag::Null; }
  Cell& operator=(Cell&& o) noexcept {
    if (this != &o) { release(); tag = o.tag; u = o.u; o.tag = Tag::Null; }
    return *this;
  }
  ~Cell() { release(); }
  void release() {
    switch (tag) {
      case Tag::String: case Tag::Bytes: case Tag::Json: free(u.s.p); break;
      case Tag::Numeric: free(u.num); break;
      case Tag::Array: delete u.arr; break;
      default: break;
    }
    tag = Tag::Null;
  }
  void set_bytes(Tag t, const char* p, size_t n) {
    release();
    tag = t;
    u.s.p = (char*)malloc(n ? n : 1);
    if (n) memcpy(u.s.p, p, n);
    u.s.len = n;
  }
  void set_numeric(const Numeric& n) {
    release();
    tag = Tag::Numeric;
    size_t nd = n.digits.size();
    NumBlock* b = (NumBlock*)malloc(sizeof(NumBlock) + (nd ? nd - 1 : 0) * sizeof(int16_t));
    b->kind = n.kind; b->sign = n.sign; b->weight = n.weight; b->scale = n.scale; b->_pad = 0;
    b->ndigits = (uint32_t)nd;
    if (nd) memcpy(b->digits, n.digits.data(), nd * 2);
    u.num = b;
  }
  sv str() const { return sv(u.s.p, u.s.len); }
  Cell clone() const;  // Cell::clone (codec/event.rs:970)
};

inline Cell Cell::clone() const {
  Cell c;
  switch (tag) {
    case Tag::String: case Tag::Bytes: case Tag::Json: c.set_bytes(tag, u.s.p, u.s.len); break;
    case Tag::Numeric: {
      size_t sz = sizeof(NumBlock) + (u.num->ndigits ? u.num->ndigits - 1 : 0) * 2;
      c.tag = tag; c.u.num = (NumBlock*)malloc(sz); memcpy(c.u.num, u.num, sz); break;
    }
    case Tag::Array: {
      c.tag = tag; c.u.arr = new Arr; c.u.arr->elem_class = u.arr->elem_class;
      for (auto& e : u.arr->elems) c.u.arr->elems.push_back(e.clone());
      break;
    }
    default: c.tag = tag; c.u = u; break;
  }
  return c;
}

struct ParseErr {
  int32_t code = ETLG_E_NONE;
  const char* detail = nullptr;
};

template <class T>
struct Res {
  bool ok = false;
  T v{};
  ParseErr e{};
  static Res Ok(T v) { Res r; r.ok = true; r.v = std::move(v); return r; }
  static Res Err(int32_t code, const char* detail = nullptr) {
    Res r; r.ok = false; r.e.code = code; r.e.detail = detail; return r;
  }
};

// ----------------------------------------------------------------- type table
// tokio_postgres `Type::from_oid` for the OIDs the codec has a dedicated arm
// for (crates/etl/src/postgres/codec/text.rs:33-141); everything else is
// either a generic `_xxx` array (-> ArrayCell::String, text.rs:146-150) or
// falls through to Cell::String (text.rs:151). Unknown OIDs are TEXT
// (crates/etl-postgres/src/type_utils.rs:9-11).
inline int32_t scalar_class_of_oid(uint32_t oid) {
  switch (oid) {
    case 16: return ETLG_TC_BOOL;
    case 21: return ETLG_TC_I16;
    case 23: return ETLG_TC_I32;
    case 20: return ETLG_TC_I64;
    case 26: return ETLG_TC_U32;
    case 700: return ETLG_TC_F32;
    case 701: return ETLG_TC_F64;
    case 1700: return ETLG_TC_NUMERIC;
    case 17: return ETLG_TC_BYTEA;
    case 1082: return ETLG_TC_DATE;
    case 1083: return ETLG_TC_TIME;
    case 1266: return ETLG_TC_TIMETZ;
    case 1114: return ETLG_TC_TIMESTAMP;
    case 1184: return ETLG_TC_TIMESTAMPTZ;
    case 2950: return ETLG_TC_UUID;
    case 114: case 3802: return ETLG_TC_JSON;
    default: return ETLG_TC_STRING;
  }
}

// Element class for array OIDs with a dedicated arm, -1 otherwise.
inline int32_t dedicated_array_elem_class(uint32_t oid) {
  switch (oid) {
    case 1000: return ETLG_TC_BOOL;        // _bool
    case 1005: return ETLG_TC_I16;         // _int2
    case 1007: return ETLG_TC_I32;         // _int4
    case 1016: return ETLG_TC_I64;         // _int8
    case 1021: return ETLG_TC_F32;         // _float4
    case 1022: return ETLG_TC_F64;         // _float8
    case 1231: return ETLG_TC_NUMERIC;     // _numeric
    case 1001: return ETLG_TC_BYTEA;       // _bytea
    case 1182: return ETLG_TC_DATE;        // _date
    case 1183: return ETLG_TC_TIME;        // _time
    case 1270: return ETLG_TC_TIMETZ;      // _timetz
    case 1115: return ETLG_TC_TIMESTAMP;   // _timestamp
    case 1185: return ETLG_TC_TIMESTAMPTZ; // _timestamptz
    case 2951: return ETLG_TC_UUID;        // _uuid
    case 199: case 3807: return ETLG_TC_JSON; // _json, _jsonb
    case 1028: return ETLG_TC_U32;         // _oid
    default: return -1;
  }
}

// Built-in array types known to postgres-types 0.2.x (`Kind::Array` and a name
// starting with '_'; is_array_type, crates/etl-postgres/src/type_utils.rs:14-18).
// int2vector (22) / oidvector (30) have array kind but no '_' name -> String.
// NOTE: list restated from the crate's generated type table (not in the
// reference tree) — parity unpinned for exotic OIDs.
inline bool is_builtin_array_oid(uint32_t oid) {
  static const uint32_t k[] = {
      143, 199, 210, 270, 272, 273, 629, 651, 719, 775, 791, 1000, 1001, 1002, 1003,
      1005, 1006, 1007, 1008, 1009, 1010, 1011, 1012, 1013, 1014, 1015, 1016, 1017,
      1018, 1019, 1020, 1021, 1022, 1027, 1028, 1034, 1040, 1041, 1115, 1182, 1183,
      1185, 1187, 1231, 1263, 1270, 1561, 1563, 2201, 2207, 2208, 2209, 2210, 2211,
      2949, 2951, 3221, 3643, 3644, 3645, 3735, 3770, 3807, 3905, 3907, 3909, 3911,
      3913, 3927, 4073, 4090, 4097, 4192, 5039, 6151, 6152, 6153, 6155, 6156, 6157};
  for (uint32_t v : k)
    if (v == oid) return true;
  return false;
}

inline int32_t class_of_oid(uint32_t oid) {
  if (is_builtin_array_oid(oid)) return ETLG_TC_ARRAY;
  return scalar_class_of_oid(oid);
}

inline int32_t array_elem_class(uint32_t oid) {
  int32_t c = dedicated_array_elem_class(oid);
  return c < 0 ? ETLG_TC_STRING : c;
}

// ---------------------------------------------------------------------- UTF-8
// core::str::from_utf8 (Rust std): strict RFC 3629 — no overlongs, no
// surrogates, max U+10FFFF. Call site: codec/event.rs:976.
inline bool utf8_valid(const u8* p, size_t n) {
  size_t i = 0;
  while (i < n) {
    u8 c = p[i];
    if (c < 0x80) { i++; continue; }
    if (c >= 0xC2 && c <= 0xDF) {
      if (i + 1 >= n || (p[i + 1] & 0xC0) != 0x80) return false;
      i += 2;
    } else if (c >= 0xE0 && c <= 0xEF) {
      if (i + 2 >= n) return false;
      u8 c1 = p[i + 1], c2 = p[i + 2];
      if ((c1 & 0xC0) != 0x80 || (c2 & 0xC0) != 0x80) return false;
      if (c == 0xE0 && c1 < 0xA0) return false;
      if (c == 0xED && c1 > 0x9F) return false;
      i += 3;
    } else if (c >= 0xF0 && c <= 0xF4) {
      if (i + 3 >= n) return false;
      u8 c1 = p[i + 1], c2 = p[i + 2], c3 = p[i + 3];
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

// Length in bytes of a Unicode White_Space char at p (0 if none): what
// str::trim / trim_end strip.
inline size_t ws_len_at(const u8* p, size_t n) {
  if (n == 0) return 0;
  u8 c = p[0];
  if ((c >= 0x09 && c <= 0x0D) || c == 0x20) return 1;
  if (c == 0xC2 && n >= 2 && (p[1] == 0x85 || p[1] == 0xA0)) return 2;
  if (c == 0xE1 && n >= 3 && p[1] == 0x9A && p[2] == 0x80) return 3;
  if (c == 0xE2 && n >= 3) {
    if (p[1] == 0x80 && ((p[2] >= 0x80 && p[2] <= 0x8A) || p[2] == 0xA8 || p[2] == 0xA9 ||
                         p[2] == 0xAF))
      return 3;
    if (p[1] == 0x81 && p[2] == 0x9F) return 3;
  }
  if (c == 0xE3 && n >= 3 && p[1] == 0x80 && p[2] == 0x80) return 3;
  return 0;
}

inline sv trim_start(sv s) {
  for (;;) {
    size_t l = ws_len_at((const u8*)s.data(), s.size());
    if (!l) return s;
    s.remove_prefix(l);
  }
}

inline sv trim_end(sv s) {
  for (;;) {
    if (s.empty()) return s;
    // find the start of the last char
    size_t i = s.size() - 1;
    while (i > 0 && (((u8)s[i]) & 0xC0) == 0x80) i--;
    size_t l = ws_len_at((const u8*)s.data() + i, s.size() - i);
    if (l != s.size() - i) return s;
    s.remove_suffix(l);
  }
}

inline sv trim(sv s) { return trim_end(trim_start(s)); }

inline bool eq_ignore_ascii_case(sv a, const char* lit) {
  size_t n = strlen(lit);
  if (a.size() != n) return false;
  for (size_t i = 0; i < n; i++) {
    char c = a[i];
    if (c >= 'A' && c <= 'Z') c = char(c + 32);
    if (c != lit[i]) return false;
  }
  return true;
}

// ----------------------------------------------------------------- bool / hex
// parse_bool, crates/etl/src/postgres/codec/bool.rs:11-19
inline Res<bool> parse_bool(sv s) {
  if (s == "t") return Res<bool>::Ok(true);
  if (s == "f") return Res<bool>::Ok(false);
  return Res<bool>::Err(ETLG_E_BOOL, "Boolean value must be 't' or 'f'");
}

// parse_bytea_hex_string, crates/etl/src/postgres/codec/hex.rs:11-52
inline Res<std::string> parse_bytea_hex(sv s) {
  using R = Res<std::string>;
  if (s.size() < 2 || s[0] != '\\' || s[1] != 'x') return R::Err(ETLG_E_BYTEA, "Missing '\\x' prefix");
  s.remove_prefix(2);
  if (s.size() % 2 != 0) return R::Err(ETLG_E_BYTEA, "Odd number of hexadecimal digits");
  std::string out;
  out.reserve(s.size() / 2);
  auto hexv = [](u8 b) -> int {
    if (b >= '0' && b <= '9') return b - '0';
    if (b >= 'a' && b <= 'f') return b - 'a' + 10;
    if (b >= 'A' && b <= 'F') return b - 'A' + 10;
    return -1;
  };
  for (size_t i = 0; i < s.size(); i += 2) {
    int h = hexv((u8)s[i]);
    if (h < 0) return R::Err(ETLG_E_BYTEA, "Invalid hexadecimal digit");
    int l = hexv((u8)s[i + 1]);
    if (l < 0) return R::Err(ETLG_E_BYTEA, "Invalid hexadecimal digit");
    out.push_back(char((h << 4) | l));
  }
  return R::Ok(std::move(out));
}

// ------------------------------------------------------------------- integers
// Rust core::num `from_str_radix(_, 10)` as used by `str::parse::<iN/uN>()`
// (call sites codec/text.rs:40-51,135-138): empty -> error; a lone "+" or "-"
// -> error; one optional leading '+' (both signednesses) or '-' (signed only;
// for unsigned '-' is an invalid digit); then ASCII digits only; exact
// overflow detection. KATs: codec/text.rs:443-483.
inline Res<int64_t> parse_int(sv s, bool is_signed, int64_t minv, uint64_t maxv) {
  using R = Res<int64_t>;
  if (s.empty()) return R::Err(ETLG_E_INT);
  bool neg = false;
  if (s[0] == '+' || s[0] == '-') {
    if (s.size() == 1) return R::Err(ETLG_E_INT);
    if (s[0] == '-') {
      if (!is_signed) return R::Err(ETLG_E_INT);
      neg = true;
    }
    s.remove_prefix(1);
  }
  // accumulate magnitude in 128-bit-safe fashion
  unsigned __int128 mag = 0;
  unsigned __int128 lim = neg ? (unsigned __int128)(-(minv + 1)) + 1 : (unsigned __int128)maxv;
  for (char ch : s) {
    if (ch < '0' || ch > '9') return R::Err(ETLG_E_INT);
    mag = mag * 10 + (unsigned)(ch - '0');
    // Pos/NegOverflow and InvalidDigit map to the same EtlError (error.rs:642-651).
    if (mag > lim) return R::Err(ETLG_E_INT);
  }
  int64_t v = neg ? (int64_t)(-(__int128)mag) : (int64_t)(uint64_t)mag;
  return R::Ok(v);
}

// --------------------------------------------------------------------- floats
// Rust core::num::dec2flt (`str::parse::<f32/f64>()`, call sites
// codec/text.rs:52-59). Grammar: [+-] ( "inf" | "infinity" | "nan" )
// case-insensitively, or  digits* [ '.' digits* ] [ (e|E) [+-] digits+ ] with
// at least one mantissa digit. Result is correctly rounded; f32 is rounded
// directly from the decimal string (no double rounding); overflow gives ±inf,
// underflow ±0 — never an error. glibc strtod/strtof are correctly rounded
// under round-to-nearest, so they are used on a pre-validated string.
// KATs: codec/text.rs:510-578.
inline bool float_grammar(sv s, bool& special, int& sp_kind, bool& neg) {
  special = false; neg = false;
  if (s.empty()) return false;
  if (s[0] == '+' || s[0] == '-') { neg = s[0] == '-'; s.remove_prefix(1); }
  if (s.empty()) return false;
  if (eq_ignore_ascii_case(s, "inf") || eq_ignore_ascii_case(s, "infinity")) {
    special = true; sp_kind = 1; return true;
  }
  if (eq_ignore_ascii_case(s, "nan")) { special = true; sp_kind = 2; return true; }
  size_t i = 0, nd = 0;
  while (i < s.size() && s[i] >= '0' && s[i] <= '9') { i++; nd++; }
  if (i < s.size() && s[i] == '.') {
    i++;
    while (i < s.size() && s[i] >= '0' && s[i] <= '9') { i++; nd++; }
  }
  if (nd == 0) return false;
  if (i < s.size() && (s[i] == 'e' || s[i] == 'E')) {
    i++;
    if (i < s.size() && (s[i] == '+' || s[i] == '-')) i++;
    size_t ne = 0;
    while (i < s.size() && s[i] >= '0' && s[i] <= '9') { i++; ne++; }
    if (ne == 0) return false;
  }
  return i == s.size();
}

inline Res<uint64_t> parse_f64_bits(sv s) {
  using R = Res<uint64_t>;
  bool special, neg; int k = 0;
  if (!float_grammar(s, special, k, neg)) return R::Err(ETLG_E_FLOAT);
  double d;
  if (special) {
    d = k == 1 ? INFINITY : NAN;
    if (neg) d = -d;
  } else {
    std::string z(s);
    d = strtod(z.c_str(), nullptr);
  }
  uint64_t bits; memcpy(&bits, &d, 8);
  return R::Ok(bits);
}

inline Res<uint64_t> parse_f32_bits(sv s) {
  using R = Res<uint64_t>;
  bool special, neg; int k = 0;
  if (!float_grammar(s, special, k, neg)) return R::Err(ETLG_E_FLOAT);
  float f;
  if (special) {
    f = k == 1 ? INFINITY : NAN;
    if (neg) f = -f;
  } else {
    std::string z(s);
    f = strtof(z.c_str(), nullptr);
  }
  uint32_t bits; memcpy(&bits, &f, 4);
  return R::Ok((uint64_t)bits);
}

// Which float texts the DEVICE decodes itself (include/etlg.h, "float cells"). This is a rule of OUR contract, not
// reference behaviour. It is STATED here independently of the device's implementation (etl_amd/csrc/float_fast.h is not
// included): a text is decoded on the device iff its value is already determined by its first 19 significant digits, i.e.
//   * zero, inf / infinity / nan, or at most 19 significant digits: always;
//   * more than 19 significant digits (a non-zero digit was cut off): iff w x 10^q and (w + 1) x 10^q — the two decimals
//     that bracket the text, w = the first 19 significant digits — round to the SAME float. Both roundings are taken with
//     glibc strtod / strtof on the synthesised decimals, exact and unrelated to the device's arithmetic.
// The device reaches the same verdict with Clinger's exact path and the Eisel-Lemire algorithm; the one place where the two
// could part is a text on which Eisel-Lemire itself is inconclusive (the device then defers although the value is
// determined): none is known, tests/native/float_rule_check.cpp compares the two verdicts on millions of texts, and a
// parity failure on such a text would name it.
// Returns 0 decode on device, 1 deferred, 2 malformed. Only used in CONTRACT mode.
inline int float_device_rule(sv s, bool is32) {
  bool special, neg; int k = 0;
  if (!float_grammar(s, special, k, neg)) return 2;
  if (special) return 0;
  size_t i = (s[0] == '+' || s[0] == '-') ? 1 : 0;
  std::string w;            // first 19 significant digits
  long long q = 0;          // decimal exponent of the last digit position read so far
  bool frac = false, cut = false;
  size_t zeros = 0;         // zeros after a significant digit, not yet appended
  for (; i < s.size(); i++) {
    const char c = s[i];
    if (c == '.') { frac = true; continue; }
    if (c < '0' || c > '9') break;
    if (frac) q--;
    if (c == '0') { if (!w.empty() || cut) zeros++; continue; }
    // a non-zero digit: the zeros held back, then the digit; whatever does not fit 19 digits only moves the exponent
    for (size_t z = 0; z <= zeros; z++) {
      const char d = z < zeros ? '0' : c;
      if (w.size() < 19 && !cut) w.push_back(d);
      else { cut = true; q++; }
    }
    zeros = 0;
  }
  q += (long long)zeros;    // trailing zeros are not part of w
  if (i < s.size()) {       // exponent (the grammar was checked above)
    i++;
    bool eneg = false;
    if (s[i] == '+' || s[i] == '-') { eneg = s[i] == '-'; i++; }
    long long ex = 0;
    for (; i < s.size(); i++) if (ex < 100000) ex = ex * 10 + (s[i] - '0');
    q += eneg ? -ex : ex;
  }
  if (w.empty()) return 0;  // zero, whatever the exponent
  if (!cut) return 0;
  // w + 1 as a decimal string
  std::string w1 = w;
  int p = (int)w1.size() - 1;
  while (p >= 0 && w1[(size_t)p] == '9') { w1[(size_t)p] = '0'; p--; }
  if (p >= 0) w1[(size_t)p]++; else w1.insert(w1.begin(), '1');
  const std::string lo = w + "e" + std::to_string(q), hi = w1 + "e" + std::to_string(q);
  if (is32) {
    const float a = strtof(lo.c_str(), nullptr), b = strtof(hi.c_str(), nullptr);
    return memcmp(&a, &b, 4) == 0 ? 0 : 1;
  }
  const double a = strtod(lo.c_str(), nullptr), b = strtod(hi.c_str(), nullptr);
  return memcmp(&a, &b, 8) == 0 ? 0 : 1;
}

// -------------------------------------------------------------------- numeric
// PgNumeric::from_str            crates/etl-postgres/src/numeric.rs:108-135
// parse_special_value            :246-267
// parse_numeric_value            :276-396
// convert_to_base_10000          :404-458
// KATs :566-953; fuzz/corpus/numeric_text_roundtrip/*.
inline Res<Numeric> numeric_convert_to_base_10000(const std::vector<u8>& dd, int32_t dweight,
                                                  uint32_t dscale, u8 sign) {
  using R = Res<Numeric>;
  if (dscale > 0xFFFF) return R::Err(ETLG_E_NUMERIC, "Value out of range");
  Numeric out;
  out.kind = ETLG_NUM_VALUE;
  out.scale = (uint16_t)dscale;
  auto zero = [&]() { out.sign = 0; out.weight = 0; out.digits.clear(); return R::Ok(out); };
  if (dd.empty()) return zero();
  int32_t weight = dweight >= 0 ? (dweight + 4) / 4 - 1 : -((-dweight - 1) / 4 + 1);
  int32_t offset = (weight + 1) * 4 - (dweight + 1);
  size_t first = dd.size(), last = 0;
  for (size_t i = 0; i < dd.size(); i++)
    if (dd[i] != 0) { first = i; break; }
  if (first == dd.size()) return zero();
  for (size_t i = dd.size(); i-- > 0;)
    if (dd[i] != 0) { last = i; break; }
  int32_t first_group = (offset + (int32_t)first) / 4;
  int32_t last_group = (offset + (int32_t)last) / 4;
  int32_t final_weight = weight - first_group;
  if (final_weight < -32768 || final_weight > 32767) return R::Err(ETLG_E_NUMERIC, "Value out of range");
  for (int32_t g = first_group; g <= last_group; g++) {
    int16_t digit = 0;
    for (int32_t pos = g * 4; pos < g * 4 + 4; pos++) {
      int64_t idx = (int64_t)pos - offset;
      u8 d = (idx >= 0 && (size_t)idx < dd.size()) ? dd[(size_t)idx] : 0;
      digit = (int16_t)(digit * 10 + d);
    }
    out.digits.push_back(digit);
  }
  out.sign = sign;
  out.weight = (int16_t)final_weight;
  return R::Ok(std::move(out));
}

inline Res<Numeric> parse_numeric_value(sv b, u8 sign) {
  using R = Res<Numeric>;
  auto at = [&](size_t i) -> int { return i < b.size() ? (u8)b[i] : -1; };
  auto isd = [](int c) { return c >= '0' && c <= '9'; };
  std::vector<u8> dd;
  bool have_dp = false;
  int32_t dweight = -1;
  uint32_t dscale = 0;
  size_t pos = 0;
  if (at(0) == '.') { have_dp = true; pos++; }
  if (!isd(at(pos))) return R::Err(ETLG_E_NUMERIC, "Invalid syntax");
  while (pos < b.size()) {
    int c = at(pos);
    if (isd(c)) {
      pos++;
      dd.push_back((u8)(c - '0'));
      if (!have_dp) dweight++; else dscale++;
    } else if (c == '.') {
      if (have_dp) return R::Err(ETLG_E_NUMERIC, "Invalid syntax");
      have_dp = true;
      pos++;
      if (at(pos) == '_') return R::Err(ETLG_E_NUMERIC, "Invalid syntax");
    } else if (c == '_') {
      pos++;
      if (!isd(at(pos))) return R::Err(ETLG_E_NUMERIC, "Invalid syntax");
    } else {
      break;
    }
  }
  if (at(pos) == 'e' || at(pos) == 'E') {
    pos++;
    int64_t exponent = 0;
    bool exp_neg = false;
    if (at(pos) == '+') pos++;
    else if (at(pos) == '-') { exp_neg = true; pos++; }
    if (!isd(at(pos))) return R::Err(ETLG_E_NUMERIC, "Invalid syntax");
    while (pos < b.size()) {
      int c = at(pos);
      if (isd(c)) {
        pos++;
        exponent = exponent * 10 + (c - '0');
        if (exponent > (int64_t)(INT32_MAX / 2)) return R::Err(ETLG_E_NUMERIC, "Value out of range");
      } else if (c == '_') {
        pos++;
        if (!isd(at(pos))) return R::Err(ETLG_E_NUMERIC, "Invalid syntax");
      } else {
        break;
      }
    }
    if (exp_neg) exponent = -exponent;
    dweight += (int32_t)exponent;
    dscale = ((int64_t)dscale - exponent) < 0 ? 0u : (uint32_t)((int64_t)dscale - exponent);
  }
  if (pos != b.size()) return R::Err(ETLG_E_NUMERIC, "Invalid syntax");
  if (dscale > 16383) return R::Err(ETLG_E_NUMERIC, "Value out of range");
  return numeric_convert_to_base_10000(dd, dweight, dscale, sign);
}

inline Res<Numeric> parse_numeric(sv input) {
  using R = Res<Numeric>;
  sv t = trim(input);
  if (t.empty()) return R::Err(ETLG_E_NUMERIC, "Invalid syntax");
  u8 sign = 0;
  bool explicit_sign = false;
  sv rest = t;
  if (t[0] == '+') { rest = t.substr(1); explicit_sign = true; }
  else if (t[0] == '-') { sign = 1; rest = t.substr(1); explicit_sign = true; }
  bool regular = !rest.empty() && ((rest[0] >= '0' && rest[0] <= '9') || rest[0] == '.');
  if (!regular) {
    sv r = trim_end(rest);
    Numeric n;
    if (eq_ignore_ascii_case(r, "nan")) {
      if (explicit_sign) return R::Err(ETLG_E_NUMERIC, "Invalid syntax");
      n.kind = ETLG_NUM_NAN;
      return R::Ok(n);
    }
    if (eq_ignore_ascii_case(r, "infinity") || eq_ignore_ascii_case(r, "inf")) {
      n.kind = sign ? ETLG_NUM_NINF : ETLG_NUM_PINF;
      return R::Ok(n);
    }
    return R::Err(ETLG_E_NUMERIC, "Invalid syntax");
  }
  return parse_numeric_value(rest, sign);
}

// ------------------------------------------------------------------- calendar
// chrono::NaiveDate::from_ymd_opt + num_days_from_ce (proleptic Gregorian,
// 0001-01-01 is day 1; year 0 exists).
inline bool is_leap(int32_t y) { return (y % 4 == 0 && y % 100 != 0) || y % 400 == 0; }
inline int32_t days_in_month(int32_t y, uint32_t m) {
  static const int k[] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
  return (m == 2 && is_leap(y)) ? 29 : k[m - 1];
}
inline int64_t days_from_civil(int64_t y, unsigned m, unsigned d) {  // days since 1970-01-01
  y -= m <= 2;
  const int64_t era = (y >= 0 ? y : y - 399) / 400;
  const unsigned yoe = (unsigned)(y - era * 400);
  const unsigned doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
  const unsigned doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  return era * 146097 + (int64_t)doe - 719468;
}
inline void civil_from_days(int64_t z, int64_t& y, unsigned& m, unsigned& d) {
  z += 719468;
  const int64_t era = (z >= 0 ? z : z - 146096) / 146097;
  const unsigned doe = (unsigned)(z - era * 146097);
  const unsigned yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  y = (int64_t)yoe + era * 400;
  const unsigned doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  const unsigned mp = (5 * doy + 2) / 153;
  d = doy - (153 * mp + 2) / 5 + 1;
  m = mp < 10 ? mp + 3 : mp - 9;
  y += m <= 2;
}
constexpr int64_t kCeToUnixDays = 719163;  // num_days_from_ce(1970-01-01)
// chrono NaiveDate year range: MIN_YEAR..=MAX_YEAR = -262143..=262142.
inline std::optional<int32_t> ymd_to_ce_days(int64_t y, uint32_t m, uint32_t d) {
  if (y < -262143 || y > 262142) return std::nullopt;
  if (m < 1 || m > 12) return std::nullopt;
  if (d < 1 || (int32_t)d > days_in_month((int32_t)y, m)) return std::nullopt;
  return (int32_t)(days_from_civil(y, m, d) + kCeToUnixDays);
}

struct TimeVal { uint32_t secs, nanos; };
struct DateTimeVal { int32_t date; uint32_t secs, nanos; };

// parse_two_digits, codec/time.rs:75-84
inline std::optional<uint32_t> two_digits(u8 hi, u8 lo) {
  u8 h = (u8)(hi - '0'), l = (u8)(lo - '0');
  if (h > 9 || l > 9) return std::nullopt;
  return (uint32_t)h * 10 + l;
}

// parse_iso_date_fast, codec/time.rs:89-100
inline std::optional<int32_t> iso_date_fast(sv b) {
  if (b.size() != 10 || b[4] != '-' || b[7] != '-') return std::nullopt;
  auto y1 = two_digits(b[0], b[1]), y2 = two_digits(b[2], b[3]);
  auto mo = two_digits(b[5], b[6]), da = two_digits(b[8], b[9]);
  if (!y1 || !y2 || !mo || !da) return std::nullopt;
  return ymd_to_ce_days((int64_t)*y1 * 100 + *y2, *mo, *da);
}

// parse_iso_time_fast, codec/time.rs:107-141 (NaiveTime::from_hms_nano_opt:
// h<24, m<60, s<60, nanos<2e9 and nanos>=1e9 only when s==59 — the fast path
// never produces nanos>=1e9).
inline std::optional<TimeVal> iso_time_fast(sv b) {
  if (b.size() < 8 || b[2] != ':' || b[5] != ':') return std::nullopt;
  auto h = two_digits(b[0], b[1]), m = two_digits(b[3], b[4]), s = two_digits(b[6], b[7]);
  if (!h || !m || !s) return std::nullopt;
  uint32_t nanos = 0;
  if (b.size() != 8) {
    if (b[8] != '.') return std::nullopt;
    sv f = b.substr(9);
    if (f.empty() || f.size() > 9) return std::nullopt;
    for (char ch : f) {
      u8 d = (u8)(ch - '0');
      if (d > 9) return std::nullopt;
      nanos = nanos * 10 + d;
    }
    for (size_t i = f.size(); i < 9; i++) nanos *= 10;
  }
  if (*h >= 24 || *m >= 60 || *s >= 60) return std::nullopt;
  return TimeVal{*h * 3600 + *m * 60 + *s, nanos};
}

// parse_iso_timestamp_fast, codec/time.rs:145-154
inline std::optional<DateTimeVal> iso_timestamp_fast(sv b) {
  if (b.size() < 19 || b[10] != ' ') return std::nullopt;
  auto d = iso_date_fast(b.substr(0, 10));
  if (!d) return std::nullopt;
  auto t = iso_time_fast(b.substr(11));
  if (!t) return std::nullopt;
  return DateTimeVal{*d, t->secs, t->nanos};
}

// ---- chrono 0.4.44 `parse_from_str` restated for the three format strings
// the reference uses (crates/etl-postgres/src/time.rs:13-21): "%Y-%m-%d",
// "%H:%M:%S%.f", "%Y-%m-%d %H:%M:%S%.f". Behaviour per chrono's
// format/parse.rs: numeric items skip leading whitespace, take 1..=width ASCII
// digits (%Y: width 4 unsigned, or any length after an explicit sign); a
// format space matches zero or more whitespace; "%.f" consumes '.' + 1..9
// digits (extra digits are skipped) or nothing; trailing input is an error;
// second == 60 is a leap second (sec 59, nanos + 1e9).
// PARITY UNPINNED beyond the KATs at codec/time.rs:181-269 — the device never
// emulates this; such shapes are handed back DEFERRED.
struct ChronoScan {
  sv s;
  bool fail = false;
  void trim_ws() { s = trim_start(s); }
  bool number(size_t minw, size_t maxw, int64_t& v) {
    size_t i = 0; v = 0;
    while (i < s.size() && i < maxw && s[i] >= '0' && s[i] <= '9') {
      if (v > (INT64_MAX - 9) / 10) return false;
      v = v * 10 + (s[i] - '0'); i++;
    }
    if (i < minw) return false;
    s.remove_prefix(i);
    return true;
  }
  bool lit(char c) {
    if (s.empty() || s[0] != c) return false;
    s.remove_prefix(1);
    return true;
  }
  bool year(int64_t& y) {
    trim_ws();
    if (!s.empty() && s[0] == '-') { s.remove_prefix(1); if (!number(1, SIZE_MAX, y)) return false; y = -y; return true; }
    if (!s.empty() && s[0] == '+') { s.remove_prefix(1); return number(1, SIZE_MAX, y); }
    return number(1, 4, y);
  }
  bool num2(int64_t& v) { trim_ws(); return number(1, 2, v); }
  bool frac(uint32_t& nanos, bool& had) {
    had = false; nanos = 0;
    if (s.empty() || s[0] != '.') return true;
    sv t = s.substr(1);
    size_t i = 0; uint32_t v = 0;
    while (i < t.size() && i < 9 && t[i] >= '0' && t[i] <= '9') { v = v * 10 + (t[i] - '0'); i++; }
    if (i == 0) return false;
    for (size_t k = i; k < 9; k++) v *= 10;
    while (i < t.size() && t[i] >= '0' && t[i] <= '9') i++;
    s = t.substr(i);
    nanos = v; had = true;
    return true;
  }
};

inline std::optional<int32_t> chrono_date(ChronoScan& c) {
  int64_t y, m, d;
  if (!c.year(y) || !c.lit('-') || !c.num2(m) || !c.lit('-') || !c.num2(d)) return std::nullopt;
  return ymd_to_ce_days(y, (uint32_t)m, (uint32_t)d);
}
inline std::optional<TimeVal> chrono_time(ChronoScan& c) {
  int64_t h, m, s; uint32_t nanos; bool had;
  if (!c.num2(h) || !c.lit(':') || !c.num2(m) || !c.lit(':') || !c.num2(s)) return std::nullopt;
  if (!c.frac(nanos, had)) return std::nullopt;
  if (h >= 24 || m >= 60 || s > 60) return std::nullopt;
  if (s == 60) { s = 59; nanos += 1000000000u; }
  return TimeVal{(uint32_t)(h * 3600 + m * 60 + s), nanos};
}

// parse_postgres_date, codec/time.rs:21-27
inline Res<int32_t> parse_pg_date(sv v) {
  if (auto d = iso_date_fast(v)) return Res<int32_t>::Ok(*d);
  ChronoScan c{v};
  auto d = chrono_date(c);
  if (!d || !c.s.empty()) return Res<int32_t>::Err(ETLG_E_DATETIME);
  return Res<int32_t>::Ok(*d);
}
// parse_postgres_time, codec/time.rs:35-41
inline Res<TimeVal> parse_pg_time(sv v) {
  if (auto t = iso_time_fast(v)) return Res<TimeVal>::Ok(*t);
  ChronoScan c{v};
  auto t = chrono_time(c);
  if (!t || !c.s.empty()) return Res<TimeVal>::Err(ETLG_E_DATETIME);
  return Res<TimeVal>::Ok(*t);
}
// parse_postgres_timestamp, codec/time.rs:49-55
inline Res<DateTimeVal> parse_pg_timestamp(sv v) {
  using R = Res<DateTimeVal>;
  if (auto t = iso_timestamp_fast(v)) return R::Ok(*t);
  ChronoScan c{v};
  auto d = chrono_date(c);
  if (!d) return R::Err(ETLG_E_DATETIME);
  c.trim_ws();  // the format's literal space
  auto t = chrono_time(c);
  if (!t || !c.s.empty()) return R::Err(ETLG_E_DATETIME);
  return R::Ok(DateTimeVal{*d, t->secs, t->nanos});
}

// parse_postgres_utc_offset, crates/etl-postgres/src/time.rs:143-207
inline std::optional<int32_t> parse_utc_offset(sv v) {
  if (v.empty()) return std::nullopt;
  int sign;
  if (v[0] == '+') sign = 1; else if (v[0] == '-') sign = -1; else return std::nullopt;
  v.remove_prefix(1);
  auto two = [](sv x) -> std::optional<int32_t> {
    if (x.size() != 2 || x[0] < '0' || x[0] > '9' || x[1] < '0' || x[1] > '9') return std::nullopt;
    return (x[0] - '0') * 10 + (x[1] - '0');
  };
  int32_t h = 0, m = 0, s = 0;
  if (v.find(':') != sv::npos) {
    std::vector<sv> parts;
    size_t st = 0;
    for (;;) {
      size_t p = v.find(':', st);
      if (p == sv::npos) { parts.push_back(v.substr(st)); break; }
      parts.push_back(v.substr(st, p - st));
      st = p + 1;
    }
    if (parts.size() < 2 || parts.size() > 3) return std::nullopt;
    auto a = two(parts[0]), b = two(parts[1]);
    if (!a || !b) return std::nullopt;
    h = *a; m = *b;
    if (parts.size() == 3) { auto c = two(parts[2]); if (!c) return std::nullopt; s = *c; }
  } else {
    if (v.size() == 2) { auto a = two(v); if (!a) return std::nullopt; h = *a; }
    else if (v.size() == 4) { auto a = two(v.substr(0, 2)), b = two(v.substr(2)); if (!a || !b) return std::nullopt; h = *a; m = *b; }
    else if (v.size() == 6) { auto a = two(v.substr(0, 2)), b = two(v.substr(2, 2)), c = two(v.substr(4)); if (!a || !b || !c) return std::nullopt; h = *a; m = *b; s = *c; }
    else return std::nullopt;
  }
  if (m >= 60 || s >= 60) return std::nullopt;
  int32_t total = h * 3600 + m * 60 + s;
  if (total >= 16 * 3600) return std::nullopt;
  return sign * total;  // FixedOffset::east_opt always succeeds below 24h
}

// split_utc_offset, crates/etl-postgres/src/time.rs:135-139 and
// split_timestamp_offset, codec/time.rs:157-161: last '+'/'-' whose byte index
// is > min_index.
inline std::optional<size_t> split_offset_index(sv v, size_t min_index) {
  for (size_t i = v.size(); i-- > 0;) {
    if (i > min_index && (v[i] == '+' || v[i] == '-')) return i;
  }
  return std::nullopt;
}

struct TimeTzVal { uint32_t secs, nanos; int32_t offset; };

// parse_postgres_timetz, crates/etl-postgres/src/time.rs:121-127 (chrono only,
// no fast path in the reference).
inline Res<TimeTzVal> parse_pg_timetz(sv v) {
  using R = Res<TimeTzVal>;
  auto idx = split_offset_index(v, 0);
  if (!idx) return R::Err(ETLG_E_DATETIME);
  sv t = trim_end(v.substr(0, *idx));
  ChronoScan c{t};
  auto tv = chrono_time(c);
  if (!tv || !c.s.empty()) return R::Err(ETLG_E_DATETIME);
  auto off = parse_utc_offset(v.substr(*idx));
  if (!off) return R::Err(ETLG_E_DATETIME);
  return R::Ok(TimeTzVal{tv->secs, tv->nanos, *off});
}

// parse_postgres_timestamptz, codec/time.rs:63-71 + `.into()` UTC
// normalisation at codec/text.rs:108-111.
inline Res<DateTimeVal> parse_pg_timestamptz(sv v) {
  using R = Res<DateTimeVal>;
  auto idx = split_offset_index(v, 10);
  if (!idx) return R::Err(ETLG_E_DATETIME);
  auto ts = parse_pg_timestamp(trim_end(v.substr(0, *idx)));
  if (!ts.ok) return R::Err(ETLG_E_DATETIME);
  auto off = parse_utc_offset(v.substr(*idx));
  if (!off) return R::Err(ETLG_E_DATETIME);
  // local - offset = UTC; chrono fails (-> InvalidSyntax) only outside the
  // NaiveDate range, unreachable for years the parsers accept except at the
  // extreme ends handled here.
  int64_t total = (int64_t)ts.v.date * 86400 + (int64_t)ts.v.secs - *off;
  int64_t days = total >= 0 ? total / 86400 : -((-total + 86399) / 86400);
  int64_t secs = total - days * 86400;
  int64_t y; unsigned mo, d;
  civil_from_days(days - kCeToUnixDays, y, mo, d);
  if (y < -262143 || y > 262142) return R::Err(ETLG_E_DATETIME);
  return R::Ok(DateTimeVal{(int32_t)days, (uint32_t)secs, ts.v.nanos});
}

// ----------------------------------------------------------------------- uuid
// uuid 1.23.1 `Uuid::parse_str` -> parser::try_parse: 32 bytes = simple hex;
// 36 = hyphenated; 38 = '{' hyphenated '}'; 45 = "urn:uuid:" hyphenated;
// anything else invalid. Hex is case-insensitive; hyphens must sit at
// 8/13/18/23. Call site codec/text.rs:117-120; KAT :780-791 and
// fuzz/corpus/parse_text_cell/uuid_scalar.
inline Res<std::string> parse_uuid(sv s) {
  using R = Res<std::string>;
  auto hexv = [](u8 b) -> int {
    if (b >= '0' && b <= '9') return b - '0';
    if (b >= 'a' && b <= 'f') return b - 'a' + 10;
    if (b >= 'A' && b <= 'F') return b - 'A' + 10;
    return -1;
  };
  std::string out(16, '\0');
  if (s.size() == 32) {
    for (int i = 0; i < 16; i++) {
      int h = hexv((u8)s[2 * i]), l = hexv((u8)s[2 * i + 1]);
      if (h < 0 || l < 0) return R::Err(ETLG_E_UUID);
      out[i] = char((h << 4) | l);
    }
    return R::Ok(out);
  }
  sv h;
  if (s.size() == 36) h = s;
  else if (s.size() == 38 && s.front() == '{' && s.back() == '}') h = s.substr(1, 36);
  else if (s.size() == 45 && s.substr(0, 9) == "urn:uuid:") h = s.substr(9);
  else return R::Err(ETLG_E_UUID);
  if (h[8] != '-' || h[13] != '-' || h[18] != '-' || h[23] != '-') return R::Err(ETLG_E_UUID);
  static const int starts[] = {0, 2, 4, 6, 9, 11, 14, 16, 19, 21, 24, 26, 28, 30, 32, 34};
  for (int i = 0; i < 16; i++) {
    int hi = hexv((u8)h[starts[i]]), lo = hexv((u8)h[starts[i] + 1]);
    if (hi < 0 || lo < 0) return R::Err(ETLG_E_UUID);
    out[i] = char((hi << 4) | lo);
  }
  return R::Ok(out);
}

// ----------------------------------------------------------------------- json
// serde_json 1.0.149 `from_str::<Value>` with `arbitrary_precision`: RFC 8259
// grammar, numbers keep their literal text (any length/exponent), recursion
// limit 128 (127 nested containers parse, the 128th is RecursionLimitExceeded), trailing non-whitespace is an error. The oracle validates and
// keeps the raw text (Value equality is delegated to the host finish step).
// Call site codec/text.rs:126-129; KATs :794-822.
struct JsonScan {
  const char* p; const char* e; int depth = 0; bool ok = true;
  void ws() { while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
  bool lit(const char* l) { size_t n = strlen(l); if ((size_t)(e - p) < n || memcmp(p, l, n)) return false; p += n; return true; }
  bool string() {
    if (p >= e || *p != '"') return false;
    p++;
    while (p < e) {
      u8 c = (u8)*p;
      if (c == '"') { p++; return true; }
      if (c < 0x20) return false;
      if (c == '\\') {
        p++;
        if (p >= e) return false;
        char x = *p;
        if (x == 'u') {
          auto hex4 = [&](unsigned& v) {
            if (e - p < 5) return false;
            v = 0;
            for (int i = 1; i <= 4; i++) {
              char h = p[i]; int d;
              if (h >= '0' && h <= '9') d = h - '0'; else if (h >= 'a' && h <= 'f') d = h - 'a' + 10; else if (h >= 'A' && h <= 'F') d = h - 'A' + 10; else return false;
              v = v * 16 + d;
            }
            p += 5;
            return true;
          };
          unsigned v;
          if (!hex4(v)) return false;
          if (v >= 0xDC00 && v <= 0xDFFF) return false;  // lone trail surrogate
          if (v >= 0xD800 && v <= 0xDBFF) {
            if (e - p < 2 || p[0] != '\\' || p[1] != 'u') return false;
            p++;
            unsigned w;
            if (!hex4(w)) return false;
            if (w < 0xDC00 || w > 0xDFFF) return false;
          }
          continue;
        }
        if (x != '"' && x != '\\' && x != '/' && x != 'b' && x != 'f' && x != 'n' && x != 'r' && x != 't') return false;
        p++;
        continue;
      }
      p++;
    }
    return false;
  }
  bool number() {
    if (p < e && *p == '-') p++;
    if (p >= e) return false;
    if (*p == '0') { p++; }
    else if (*p >= '1' && *p <= '9') { while (p < e && *p >= '0' && *p <= '9') p++; }
    else return false;
    if (p < e && *p == '.') { p++; if (p >= e || *p < '0' || *p > '9') return false; while (p < e && *p >= '0' && *p <= '9') p++; }
    if (p < e && (*p == 'e' || *p == 'E')) { p++; if (p < e && (*p == '+' || *p == '-')) p++; if (p >= e || *p < '0' || *p > '9') return false; while (p < e && *p >= '0' && *p <= '9') p++; }
    return true;
  }
  bool value() {
    ws();
    if (p >= e) return false;
    char c = *p;
    if (c == '{') {
      if (++depth >= 128) return false;  // Deserializer::remaining_depth starts at 128; entering a container that takes it to 0 fails
      p++; ws();
      if (p < e && *p == '}') { p++; depth--; return true; }
      for (;;) {
        ws();
        if (!string()) return false;
        ws();
        if (p >= e || *p != ':') return false;
        p++;
        if (!value()) return false;
        ws();
        if (p < e && *p == ',') { p++; continue; }
        if (p < e && *p == '}') { p++; depth--; return true; }
        return false;
      }
    }
    if (c == '[') {
      if (++depth >= 128) return false;
      p++; ws();
      if (p < e && *p == ']') { p++; depth--; return true; }
      for (;;) {
        if (!value()) return false;
        ws();
        if (p < e && *p == ',') { p++; continue; }
        if (p < e && *p == ']') { p++; depth--; return true; }
        return false;
      }
    }
    if (c == '"') return string();
    if (c == 't') return lit("true");
    if (c == 'f') return lit("false");
    if (c == 'n') return lit("null");
    return number();
  }
};
inline bool json_valid(sv s) {
  JsonScan j{s.data(), s.data() + s.size()};
  if (!j.value()) return false;
  j.ws();
  return j.p == j.e;
}

// ---------------------------------------------------------------- text → Cell
inline Res<Cell> parse_scalar_text(int32_t cls, sv str);

// strip_array_dimensions_prefix, codec/text.rs:163-214
inline Res<sv> strip_array_dims(sv in) {
  using R = Res<sv>;
  auto at = [&](size_t i) -> int { return i < in.size() ? (u8)in[i] : -1; };
  if (at(0) != '[') return R::Ok(in);
  auto skip_int = [&](size_t idx) -> std::optional<size_t> {
    if (at(idx) == '-') idx++;
    size_t st = idx;
    while (at(idx) >= '0' && at(idx) <= '9') idx++;
    if (idx > st) return idx;
    return std::nullopt;
  };
  size_t groups = 0, idx = 0;
  while (at(idx) == '[') {
    auto al = skip_int(idx + 1);
    if (!al) return R::Err(ETLG_E_ARRAY_DIMS);
    if (at(*al) != ':') return R::Err(ETLG_E_ARRAY_DIMS);
    auto au = skip_int(*al + 1);
    if (!au) return R::Err(ETLG_E_ARRAY_DIMS);
    if (at(*au) != ']') return R::Err(ETLG_E_ARRAY_DIMS);
    idx = *au + 1;
    groups++;
  }
  if (at(idx) != '=') return R::Err(ETLG_E_ARRAY_DIMS);
  if (groups > 1) return R::Err(ETLG_E_ARRAY_MULTIDIM);
  return R::Ok(in.substr(idx + 1));
}

// parse_cell_from_postgres_text_array, codec/text.rs:228-312. The reference
// iterates `chars()`; every structural character is ASCII and the input is
// valid UTF-8, so byte iteration is equivalent.
inline Res<Cell> parse_array_text(int32_t elem_class, sv str) {
  using R = Res<Cell>;
  auto st = strip_array_dims(str);
  if (!st.ok) return R::Err(st.e.code);
  str = st.v;
  if (str.size() < 2) return R::Err(ETLG_E_ARRAY_SHORT);
  if (str.front() != '{' || str.back() != '}') return R::Err(ETLG_E_ARRAY_BRACES);
  Cell out;
  out.tag = Tag::Array;
  out.u.arr = new Arr;
  out.u.arr->elem_class = elem_class;
  sv body = str.substr(1, str.size() - 2);
  std::string val;
  bool in_quotes = false, in_escape = false, val_quoted = false;
  size_t pos = 0;
  bool done = body.empty();
  while (!done) {
    for (;;) {
      if (pos >= body.size()) { done = true; break; }
      char c = body[pos++];
      if (in_escape) { val.push_back(c); in_escape = false; }
      else if (c == '"') { if (!in_quotes) val_quoted = true; in_quotes = !in_quotes; }
      else if (c == '\\') in_escape = true;
      else if ((c == '{' || c == '}') && !in_quotes) return R::Err(ETLG_E_ARRAY_MULTIDIM);
      else if (c == ',' && !in_quotes) break;
      else val.push_back(c);
    }
    if (in_quotes) return R::Err(ETLG_E_ARRAY_QUOTE);
    if (in_escape) return R::Err(ETLG_E_ARRAY_ESCAPE);
    if (!val_quoted && eq_ignore_ascii_case(val, "null")) {
      out.u.arr->elems.emplace_back();
    } else {
      auto e = parse_scalar_text(elem_class, val);
      if (!e.ok) return R::Err(e.e.code, e.e.detail);
      out.u.arr->elems.push_back(std::move(e.v));
    }
    val.clear();
    val_quoted = false;
  }
  return R::Ok(std::move(out));
}

// The scalar arms of parse_cell_from_postgres_text, codec/text.rs:32-153.
inline Res<Cell> parse_scalar_text(int32_t cls, sv str) {
  using R = Res<Cell>;
  Cell c;
  switch (cls) {
    case ETLG_TC_BOOL: {
      auto r = parse_bool(str);
      if (!r.ok) return R::Err(r.e.code, r.e.detail);
      c.tag = Tag::Bool; c.u.b = r.v; return R::Ok(std::move(c));
    }
    case ETLG_TC_I16: {
      auto r = parse_int(str, true, INT16_MIN, INT16_MAX);
      if (!r.ok) return R::Err(r.e.code);
      c.tag = Tag::I16; c.u.i = r.v; return R::Ok(std::move(c));
    }
    case ETLG_TC_I32: {
      auto r = parse_int(str, true, INT32_MIN, INT32_MAX);
      if (!r.ok) return R::Err(r.e.code);
      c.tag = Tag::I32; c.u.i = r.v; return R::Ok(std::move(c));
    }
    case ETLG_TC_I64: {
      auto r = parse_int(str, true, INT64_MIN, INT64_MAX);
      if (!r.ok) return R::Err(r.e.code);
      c.tag = Tag::I64; c.u.i = r.v; return R::Ok(std::move(c));
    }
    case ETLG_TC_U32: {
      auto r = parse_int(str, false, 0, UINT32_MAX);
      if (!r.ok) return R::Err(r.e.code);
      c.tag = Tag::U32; c.u.i = r.v; return R::Ok(std::move(c));
    }
    case ETLG_TC_F32: {
      auto r = parse_f32_bits(str);
      if (!r.ok) return R::Err(r.e.code);
      c.tag = Tag::F32; c.u.fbits = r.v; return R::Ok(std::move(c));
    }
    case ETLG_TC_F64: {
      auto r = parse_f64_bits(str);
      if (!r.ok) return R::Err(r.e.code);
      c.tag = Tag::F64; c.u.fbits = r.v; return R::Ok(std::move(c));
    }
    case ETLG_TC_NUMERIC: {
      auto r = parse_numeric(str);
      if (!r.ok) return R::Err(r.e.code, r.e.detail);
      c.set_numeric(r.v); return R::Ok(std::move(c));
    }
    case ETLG_TC_BYTEA: {
      auto r = parse_bytea_hex(str);
      if (!r.ok) return R::Err(r.e.code, r.e.detail);
      c.set_bytes(Tag::Bytes, r.v.data(), r.v.size()); return R::Ok(std::move(c));
    }
    case ETLG_TC_DATE: {
      auto r = parse_pg_date(str);
      if (!r.ok) return R::Err(r.e.code);
      c.tag = Tag::Date; c.u.t.date = r.v; return R::Ok(std::move(c));
    }
    case ETLG_TC_TIME: {
      auto r = parse_pg_time(str);
      if (!r.ok) return R::Err(r.e.code);
      c.tag = Tag::Time; c.u.t.secs = r.v.secs; c.u.t.nanos = r.v.nanos; return R::Ok(std::move(c));
    }
    case ETLG_TC_TIMETZ: {
      auto r = parse_pg_timetz(str);
      if (!r.ok) return R::Err(r.e.code);
      c.tag = Tag::TimeTz; c.u.t.secs = r.v.secs; c.u.t.nanos = r.v.nanos; c.u.t.offset = r.v.offset; return R::Ok(std::move(c));
    }
    case ETLG_TC_TIMESTAMP: {
      auto r = parse_pg_timestamp(str);
      if (!r.ok) return R::Err(r.e.code);
      c.tag = Tag::Timestamp; c.u.t.date = r.v.date; c.u.t.secs = r.v.secs; c.u.t.nanos = r.v.nanos; return R::Ok(std::move(c));
    }
    case ETLG_TC_TIMESTAMPTZ: {
      auto r = parse_pg_timestamptz(str);
      if (!r.ok) return R::Err(r.e.code);
      c.tag = Tag::TimestampTz; c.u.t.date = r.v.date; c.u.t.secs = r.v.secs; c.u.t.nanos = r.v.nanos; return R::Ok(std::move(c));
    }
    case ETLG_TC_UUID: {
      auto r = parse_uuid(str);
      if (!r.ok) return R::Err(r.e.code);
      c.tag = Tag::Uuid; memcpy(c.u.uuid, r.v.data(), 16); return R::Ok(std::move(c));
    }
    case ETLG_TC_JSON: {
      if (!json_valid(str)) return R::Err(ETLG_E_JSON);
      c.set_bytes(Tag::Json, str.data(), str.size()); return R::Ok(std::move(c));
    }
    default:
      c.set_bytes(Tag::String, str.data(), str.size()); return R::Ok(std::move(c));
  }
}

// parse_cell_from_postgres_text, codec/text.rs:32-153 (type switch by OID).
inline Res<Cell> parse_cell_text(uint32_t type_oid, sv str) {
  int32_t cls = class_of_oid(type_oid);
  if (cls == ETLG_TC_ARRAY) return parse_array_text(array_elem_class(type_oid), str);
  return parse_scalar_text(cls, str);
}

}  // namespace orc
