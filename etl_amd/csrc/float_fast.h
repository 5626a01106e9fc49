// float4 / float8 text -> IEEE bits: the exact fast path shared by the kernels (codec.hip.h) and a
// host-side unit test (tests/test_float_fast.py compiles this header with g++ and checks it against
// strtod / strtof on millions of texts).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define ETLG_HD __host__ __device__ __forceinline__
#else
#define ETLG_HD static inline
#endif

namespace etlg {

ETLG_HD uint32_t flt_lower(uint32_t c) { return (c - 'A' < 26u) ? c + 32 : c; }

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
ETLG_HD int parse_float_fast_t(At at, uint32_t n, bool is32, uint64_t& out) {
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
  uint32_t ndig = 0;
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
    if (nsig + add > 19) too_long = true;
    else {
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
  if (nsig == 0) {  // zero, whatever the exponent
    out = is32 ? (neg ? 0x80000000ull : 0ull) : (neg ? 0x8000000000000000ull : 0ull);
    return 0;
  }
  if (too_long || w > (1ull << 53) || q < -22 || q > 22) return 1;
  const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  const double v = q < 0 ? (double)w / p10[-q] : (double)w * p10[q];
  uint64_t bits; __builtin_memcpy(&bits, &v, 8);
  if (!is32) { out = bits | (neg ? 0x8000000000000000ull : 0ull); return 0; }
  // f64 -> f32, round to nearest even in integer arithmetic (no dependence on the denormal mode)
  if ((bits & 0x1FFFFFFFull) == 0x10000000ull) return 1;  // exactly on a float midpoint: the double rounding may be wrong
  const int32_t e = (int32_t)((bits >> 52) & 0x7FF) - 1023;  // v is a normal double >= 1e-22
  const uint64_t m = (bits & 0xFFFFFFFFFFFFFull) | (1ull << 52);
  uint32_t f;
  if (e > 127) f = 0x7F800000u;
  else if (e >= -126) {
    uint64_t r = m >> 29;                       // 24 bits
    const uint64_t rem = m & 0x1FFFFFFFull;
    if (rem > 0x10000000ull || (rem == 0x10000000ull && (r & 1))) r++;
    f = (uint32_t)(((uint64_t)(e + 127) << 23) + (r - (1ull << 23)));  // a carry out of the mantissa bumps the exponent (up to inf)
  } else {
    // float subnormal (cannot happen for |q| <= 22 and w >= 1: v >= 1e-22 > 2^-126 * 2^-23) — defer to be safe
    return 1;
  }
  out = f | (neg ? 0x80000000u : 0u);
  return 0;
}

}  // namespace etlg
