// float4 / float8 text -> IEEE bits: the exact fast path shared by the kernels (codec.hip.h) and a
// host-side unit test (tests/test_float_fast.py compiles this header with g++ and checks it against
// strtod / strtof on millions of texts).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define ETLG_FD __device__ __forceinline__
#define ETLG_TABLE __device__ const
#else
#define ETLG_FD static inline
#define ETLG_TABLE static const
#endif
#include "pow5_table.h"

namespace etlg {

ETLG_FD uint32_t flt_lower(uint32_t c) { return (c - 'A' < 26u) ? c + 32 : c; }

// ---- Eisel-Lemire (D. Lemire, "Number parsing at a gigabyte per second", SPE 2021; the algorithm Rust's
// dec2flt runs after its own fast path): w * 10^q -> the correctly rounded binary float from ONE 64 x 128-bit
// multiplication by a truncated power of five, or "inconclusive" (then the text is DEFERRED).
struct FltFormat { int mant_bits, min_exp, inf_power, rte_min, rte_max, smallest_p10, largest_p10; };
ETLG_FD FltFormat flt_format(bool is32) {
  return is32 ? FltFormat{23, -127, 0xFF, -17, 10, -64, 38} : FltFormat{52, -1023, 0x7FF, -4, 23, -342, 308};
}
ETLG_FD void mul64x64(uint64_t a, uint64_t b, uint64_t& lo, uint64_t& hi) {
  const uint64_t a0 = (uint32_t)a, a1 = a >> 32, b0 = (uint32_t)b, b1 = b >> 32;
  const uint64_t p00 = a0 * b0, p01 = a0 * b1, p10 = a1 * b0, p11 = a1 * b1;
  const uint64_t mid = (p00 >> 32) + (uint32_t)p01 + (uint32_t)p10;
  lo = (mid << 32) | (uint32_t)p00;
  hi = p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
}
// Returns 0 and (mantissa without the hidden bit, biased exponent), or 1 when the product is inconclusive.
ETLG_FD int eisel_lemire(int64_t q, uint64_t w, bool is32, uint64_t& mant, int32_t& pw2) {
  const FltFormat F = flt_format(is32);
  mant = 0; pw2 = 0;
  if (w == 0 || q < F.smallest_p10) return 0;                    // underflows to zero
  if (q > F.largest_p10) { pw2 = F.inf_power; return 0; }        // overflows to infinity
  const int lz = __builtin_clzll(w);
  w <<= lz;
  // compute_product_approx: w * (high word of 5^q), refined with the low word when the upper bits are all ones
  const int precision = F.mant_bits + 3;
  const uint64_t mask = ~0ull >> precision;
  const uint64_t p5hi = kPow5[q - kPow5Smallest][0], p5lo = kPow5[q - kPow5Smallest][1];
  uint64_t lo, hi;
  mul64x64(w, p5hi, lo, hi);
  if ((hi & mask) == mask) {
    uint64_t lo2, hi2;
    mul64x64(w, p5lo, lo2, hi2);
    lo += hi2;
    if (hi2 > lo) hi++;
  }
  if (lo == ~0ull && !(q >= -27 && q <= 55)) return 1;           // cannot tell which side of a rounding boundary
  const int upperbit = (int)(hi >> 63);
  uint64_t m = hi >> (upperbit + 64 - F.mant_bits - 3);
  int32_t p2 = (int32_t)((((152170 + 65536) * q) >> 16) + 63) + upperbit - lz - F.min_exp;
  if (p2 <= 0) {                                                  // subnormal result
    if (-p2 + 1 >= 64) return 0;
    m >>= -p2 + 1;
    m += m & 1;
    m >>= 1;
    pw2 = m < (1ull << F.mant_bits) ? 0 : 1;
    mant = m & ~(1ull << F.mant_bits);
    return 0;
  }
  // exactly half way between two floats and the product is exact: round to even
  if (lo <= 1 && q >= F.rte_min && q <= F.rte_max && (m & 3) == 1 && (m << (upperbit + 64 - F.mant_bits - 3)) == hi) m &= ~1ull;
  m += m & 1;
  m >>= 1;
  if (m >= (2ull << F.mant_bits)) { m = 1ull << F.mant_bits; p2++; }
  m &= ~(1ull << F.mant_bits);
  if (p2 >= F.inf_power) { mant = 0; pw2 = F.inf_power; return 0; }
  mant = m; pw2 = p2;
  return 0;
}

// The value of (sign) w * 10^q, w = the first (up to 19) significant digits of the mantissa, `dropped` more behind them when
// the mantissa was longer (too_long): shared by the two front ends below. Returns 0 value (bits in out), 1 defer.
ETLG_FD int flt_finish(bool neg, uint64_t w, uint32_t nsig, bool too_long, uint32_t dropped, int32_t q, bool is32, uint64_t& out) {
  if (nsig == 0) {  // zero, whatever the exponent
    out = is32 ? (neg ? 0x80000000ull : 0ull) : (neg ? 0x8000000000000000ull : 0ull);
    return 0;
  }
  if (!too_long && w <= (1ull << 53) && q >= -22 && q <= 22) {
    // Clinger's fast path: both factors are exact doubles, one correctly rounded operation
    const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                            1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const double v = q < 0 ? (double)w / p10[-q] : (double)w * p10[q];
    uint64_t bits; __builtin_memcpy(&bits, &v, 8);
    if (!is32) { out = bits | (neg ? 0x8000000000000000ull : 0ull); return 0; }
    // f32 from the double: only safe when the double is not exactly on a float midpoint; otherwise Eisel-Lemire decides
    if ((bits & 0x1FFFFFFFull) != 0x10000000ull) {
      const int32_t e = (int32_t)((bits >> 52) & 0x7FF) - 1023;  // v is a normal double >= 1e-22
      const uint64_t m = (bits & 0xFFFFFFFFFFFFFull) | (1ull << 52);
      if (e > 127) { out = 0x7F800000u | (neg ? 0x80000000u : 0u); return 0; }
      if (e >= -126) {
        uint64_t r = m >> 29;  // 24 bits, round to nearest even in integer arithmetic (no dependence on the denormal mode)
        const uint64_t rem = m & 0x1FFFFFFFull;
        if (rem > 0x10000000ull || (rem == 0x10000000ull && (r & 1))) r++;
        out = (uint32_t)(((uint64_t)(e + 127) << 23) + (r - (1ull << 23))) | (neg ? 0x80000000u : 0u);  // a carry bumps the exponent
        return 0;
      }
    }
  }
  // Eisel-Lemire on the (up to 19-digit) mantissa; a longer mantissa was truncated, so w and w + 1 bracket the
  // value and must round to the same float
  const int64_t qe = (int64_t)q + dropped;
  uint64_t m1; int32_t e1;
  if (eisel_lemire(qe, w, is32, m1, e1)) return 1;
  if (too_long) {
    uint64_t m2; int32_t e2;
    if (eisel_lemire(qe, w + 1, is32, m2, e2) || m1 != m2 || e1 != e2) return 1;
  }
  if (is32) out = (uint32_t)m1 | ((uint32_t)e1 << 23) | (neg ? 0x80000000u : 0u);
  else out = m1 | ((uint64_t)(uint32_t)e1 << 52) | (neg ? 0x8000000000000000ull : 0ull);
  return 0;
}

// f32 / f64 `str::parse` (Rust core::num::dec2flt; call sites codec/text.rs:52-59), the part that is
// exact with one IEEE operation (W. Clinger's fast path): the text is  [+-] digits [. digits] [e[+-]digits]
// or inf / infinity / nan in any case; with the mantissa digits read as an integer w (leading and
// trailing zeros dropped, the latter folded into the exponent q) the value is w * 10^q, and when
// w <= 2^53 and |q| <= 22 both w and 10^|q| are exact doubles, so ONE correctly rounded multiply or
// divide is the correctly rounded result. f32 rounds that double once more, which is only unsafe
// when the double sits exactly on the midpoint of two floats — those, every longer mantissa and
// every larger exponent are handed back DEFERRED (include/etlg.h), never approximated.
// Returns 0 value (bits in out), 1 defer, 2 malformed (ETLG_E_FLOAT).
template <class At>  // At: uint32_t operator()(uint32_t i) -> byte i of the text
ETLG_FD int parse_float_fast_t(At at, uint32_t n, bool is32, uint64_t& out) {
  uint32_t i = 0;
  bool neg = false;
  if (n && (at(0) == '+' || at(0) == '-')) { neg = at(0) == '-'; i = 1; }
  if (i >= n) return 2;
  const uint32_t c0 = flt_lower(at(i));
  if (c0 == 'i' || c0 == 'n') {  // inf | infinity | nan
    const uint32_t r = n - i;
    const char* lit = c0 == 'n' ? "nan" : (r == 3 ? "inf" : "infinity");
    const uint32_t ll = c0 == 'n' ? 3u : (r == 3 ? 3u : 8u);
    if (r != ll) return 2;
    for (uint32_t k = 0; k < ll; k++) if (flt_lower(at(i + k)) != (uint32_t)lit[k]) return 2;
    const uint64_t sign64 = neg ? 0x8000000000000000ull : 0ull;
    if (is32) out = (c0 == 'n' ? 0x7FC00000u : 0x7F800000u) | (neg ? 0x80000000u : 0u);
    else out = (c0 == 'n' ? 0x7FF8000000000000ull : 0x7FF0000000000000ull) | sign64;
    return 0;
  }
  // mantissa
  uint64_t w = 0;
  uint32_t nsig = 0;       // significant digits accumulated into w (after leading zeros)
  int32_t q = 0;           // decimal exponent of w's last digit
  uint32_t pending0 = 0;   // zeros seen after a significant digit, not yet appended (dropped if trailing)
  uint32_t ndig = 0, dropped = 0;  // dropped: significant digit positions beyond the 19 that fit w
  bool frac = false, too_long = false;
  for (; i < n; i++) {
    const uint32_t c = at(i);
    if (c == '.') { if (frac) return 2; frac = true; continue; }
    const uint32_t d = c - '0';
    if (d > 9) break;
    ndig++;
    if (frac) q--;
    if (d == 0) { if (nsig) pending0++; continue; }  // leading zeros carry nothing
    // a non-zero digit: first append the zeros held back, then the digit
    const uint32_t add = pending0 + 1;
    if (too_long || nsig + add > 19) {
      // keep what fits: the zeros held back, then the digit; positions that do not fit only move the exponent
      uint32_t room = too_long ? 0u : 19u - nsig;
      for (uint32_t z = 0; z < pending0; z++) { if (room) { w *= 10; room--; } else dropped++; }
      if (room) w = w * 10 + d; else dropped++;
      too_long = true;
    } else {
      for (uint32_t z = 0; z < pending0; z++) w *= 10;
      w = w * 10 + d;
    }
    nsig += add; pending0 = 0;
  }
  if (ndig == 0) return 2;
  // q currently counts fraction digits of ALL digits read; the zeros held back at the end are not in w
  q += (int32_t)pending0;
  if (i < n) {
    const uint32_t c = at(i);
    if (c != 'e' && c != 'E') return 2;
    i++;
    bool eneg = false;
    if (i < n && (at(i) == '+' || at(i) == '-')) { eneg = at(i) == '-'; i++; }
    if (i >= n) return 2;
    uint32_t ex = 0;
    for (; i < n; i++) {
      const uint32_t d = at(i) - '0';
      if (d > 9) return 2;
      if (ex < 100000u) ex = ex * 10 + d;
    }
    q += eneg ? -(int32_t)ex : (int32_t)ex;
  }
  return flt_finish(neg, w, nsig, too_long, dropped, q, is32, out);
}

// ---- the same grammar without the per-character loop, for texts of up to 24 bytes held in three registers (x0 = bytes 0..7,
// little endian; bytes past n may hold anything). One wave parses 64 cells in lock step, so the loop above costs every lane the
// iterations of the longest text and both sides of each of its branches (~37 k cycles per column and tile measured in k_cells'
// sizing pass, profiles/r03ad_copy_direct.txt); here the positions of the dot, the exponent and the significant digits come
// from byte-class masks, and the digits are converted eight at a time. Returns what parse_float_fast_t returns for the same text,
// or 3 = "not taken" (longer texts, inf / nan, exponents of more than five digits, non-ASCII bytes): the caller then runs the loop.
ETLG_FD uint64_t flt_eq8(uint64_t x, uint64_t c) {  // bit 7 of every byte equal to c (exact)
  const uint64_t z = x ^ (c * 0x0101010101010101ull);
  return ~(((z & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | z) & 0x8080808080808080ull;
}
ETLG_FD uint32_t flt_pack8(uint64_t m) { return (uint32_t)((m * 0x0002040810204081ull) >> 56); }  // bit 7 of byte j -> bit j
ETLG_FD uint64_t flt_digits8(uint64_t v) {  // eight ASCII digits, first character in the lowest byte, -> their value
  v = (v & 0x0F0F0F0F0F0F0F0Full) * 2561 >> 8;
  v = (v & 0x00FF00FF00FF00FFull) * 6553601 >> 16;
  return (v & 0x0000FFFF0000FFFFull) * 42949672960001ull >> 32;
}
ETLG_FD int parse_float_swar(uint64_t x0, uint64_t x1, uint64_t x2, uint32_t n, bool is32, uint64_t& out) {
  if (n == 0) return 2;
  if (n > 24) return 3;
  auto keep = [](uint64_t x, uint32_t cnt) -> uint64_t { return cnt >= 8 ? x : cnt == 0 ? 0ull : x & (~0ull >> (64 - 8 * cnt)); };
  x0 = keep(x0, n); x1 = keep(x1, n > 8 ? n - 8 : 0u); x2 = keep(x2, n > 16 ? n - 16 : 0u);
  if ((x0 | x1 | x2) & 0x8080808080808080ull) return 3;
  auto at = [&](uint32_t p) -> uint32_t { const uint64_t wd = p < 8 ? x0 : p < 16 ? x1 : x2; return (uint32_t)(wd >> (8 * (p & 7u))) & 0xFFu; };  // p < 24
  // byte classes (every byte is < 0x80 here; bytes past the text are 0, i.e. "not a digit")
  auto nondigit = [](uint64_t x) -> uint64_t { return ((x + 0x4646464646464646ull) | ~((x | 0x8080808080808080ull) - 0x3030303030303030ull)) & 0x8080808080808080ull; };
  const uint32_t ND = flt_pack8(nondigit(x0)) | (flt_pack8(nondigit(x1)) << 8) | (flt_pack8(nondigit(x2)) << 16) | 0xFF000000u;
  const uint32_t DOT = flt_pack8(flt_eq8(x0, '.')) | (flt_pack8(flt_eq8(x1, '.')) << 8) | (flt_pack8(flt_eq8(x2, '.')) << 16);
  const uint64_t lc = 0x2020202020202020ull;
  const uint32_t EXP = flt_pack8(flt_eq8(x0 | lc, 'e')) | (flt_pack8(flt_eq8(x1 | lc, 'e')) << 8) | (flt_pack8(flt_eq8(x2 | lc, 'e')) << 16);
  const uint32_t c0 = (uint32_t)x0 & 0xFFu;
  const bool neg = c0 == '-';
  const uint32_t i = (c0 == '+' || c0 == '-') ? 1u : 0u;
  if (i >= n) return 2;
  { const uint32_t ci = flt_lower(at(i)); if (ci == 'i' || ci == 'n') return 3; }
  const uint32_t p1 = i + (uint32_t)__builtin_ctz(ND >> i);              // end of the integer digits
  const bool hasdot = ((DOT >> p1) & 1u) != 0;
  const uint32_t p2 = hasdot ? p1 + 1 + (uint32_t)__builtin_ctz(ND >> (p1 + 1)) : p1;   // end of the mantissa
  const uint32_t nfrac = hasdot ? p2 - p1 - 1 : 0u, ndig = (p1 - i) + nfrac;
  if (ndig == 0) return 2;
  int32_t ex = 0;
  if (p2 < n) {
    if (!((EXP >> p2) & 1u)) return 2;
    uint32_t q1 = p2 + 1;
    bool eneg = false;
    if (q1 < n) { const uint32_t cs = at(q1); if (cs == '+' || cs == '-') { eneg = cs == '-'; q1++; } }
    if (q1 >= n) return 2;
    const uint32_t q2 = q1 + (uint32_t)__builtin_ctz(ND >> q1);
    if (q2 != n) return 2;
    if (q2 - q1 > 5) return 3;
    for (uint32_t t = q1; t < q2; t++) ex = ex * 10 + (int32_t)(at(t) - '0');
    if (eneg) ex = -ex;
  }
  // the mantissa's digits as one string: the dot goes (bytes above it move down by one)
  uint64_t y0 = x0, y1 = x1, y2 = x2;
  if (hasdot) {
    auto low = [](uint32_t cnt) -> uint64_t { return cnt >= 8 ? ~0ull : cnt == 0 ? 0ull : ~0ull >> (64 - 8 * cnt); };   // the lowest cnt bytes
    const uint64_t l0 = low(p1), l1 = low(p1 > 8 ? p1 - 8 : 0u), l2 = low(p1 > 16 ? p1 - 16 : 0u);
    const uint64_t s0 = (x0 >> 8) | (x1 << 56), s1 = (x1 >> 8) | (x2 << 56), s2 = x2 >> 8;
    y0 = (x0 & l0) | (s0 & ~l0); y1 = (x1 & l1) | (s1 & ~l1); y2 = (x2 & l2) | (s2 & ~l2);
  }
  // significant digits: from the first to the last non-zero one
  const uint32_t Z = flt_pack8(flt_eq8(y0, '0')) | (flt_pack8(flt_eq8(y1, '0')) << 8) | (flt_pack8(flt_eq8(y2, '0')) << 16);
  const uint32_t R = ((1u << ndig) - 1u) << i;   // ndig + i <= 24
  const uint32_t NZ = ~Z & R;
  const int32_t qx = ex - (int32_t)nfrac;
  if (NZ == 0) return flt_finish(neg, 0, 0, false, 0, qx, is32, out);
  const uint32_t first = (uint32_t)__builtin_ctz(NZ), last = 31u - (uint32_t)__builtin_clz(NZ);
  const uint32_t nsig = last - first + 1;
  const int32_t q = qx + (int32_t)((i + ndig - 1) - last);   // trailing zeros fold into the exponent
  const bool too_long = nsig > 19;
  const uint32_t take = too_long ? 19u : nsig, dropped = nsig - take;
  // bring the first significant digit to byte 0
  {
    const uint32_t wsft = first >> 3, bsft = 8 * (first & 7u);
    const uint64_t a0 = wsft == 0 ? y0 : wsft == 1 ? y1 : y2, a1 = wsft == 0 ? y1 : wsft == 1 ? y2 : 0ull, a2 = wsft == 0 ? y2 : 0ull;
    y0 = bsft ? (a0 >> bsft) | (a1 << (64 - bsft)) : a0;
    y1 = bsft ? (a1 >> bsft) | (a2 << (64 - bsft)) : a1;
    y2 = bsft ? a2 >> bsft : a2;
  }
  const uint64_t p10[9] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull};
  const uint32_t c1 = take < 8 ? take : 8u, c2 = take - c1 < 8 ? take - c1 : 8u, c3 = take - c1 - c2;   // c3 <= 3
  uint64_t w = flt_digits8(c1 == 8 ? y0 : y0 << (8 * (8 - c1)));
  if (c2) w = w * p10[c2] + flt_digits8(c2 == 8 ? y1 : y1 << (8 * (8 - c2)));
  if (c3) w = w * p10[c3] + flt_digits8(y2 << (8 * (8 - c3)));
  return flt_finish(neg, w, nsig, too_long, dropped, q, is32, out);
}

}  // namespace etlg
